"""Developer tool: build tuning variants of libws_stereo.so (X, ND, MAXT macros) into build/variants/
and, on the GPU box, time bench.py with each of them.
  python tools/variants.py build "8,8,768" "8,8,512" "8,4,1024"
  python tools/variants.py run config2 config3        (on the GPU box)
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VDIR = os.path.join(ROOT, "gpurun_variants")
CSRC = os.path.join(ROOT, "stereo_reconstruction_amd", "csrc")

def build(specs):
    os.makedirs(VDIR, exist_ok=True)
    for spec in specs:
        parts = spec.split(",")
        x, nd, maxt = parts[:3]
        extra = [e for e in parts[3:] if not e.startswith("-")]
        flags = [f for e in parts[3:] if e.startswith("-") for f in e.split(" ")]   # e.g. "-mllvm -amdgpu-sched-strategy=max-ilp"
        out = os.path.join(VDIR, "libws_%s.so" % "".join(ch if ch.isalnum() else "_" for ch in spec))
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-DWS_X=" + x, "-DWS_ND=" + nd, "-DWS_MAXT=" + maxt] + ["-D" + e for e in extra] + flags + [
               "-Rpass-analysis=kernel-resource-usage", "-o", out,
               ] + [os.path.join(CSRC, f) for f in ("ws_march.hip", "ws_march_nd4.hip", "ws_prepass.hip", "ws_border.hip", "ws_smooth.hip",
                                                     "ws_consumers.hip", "ws_capi.cpp")]
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            print(spec, "BUILD FAILED", r.stderr[-2000:]); continue
        info = []
        lines = r.stderr.split("\n")
        for i, l in enumerate(lines):
            if "Function Name" in l and "Li7ELi7E" in l and "march" in l:
                blk = " ".join(lines[i:i + 12])
                import re
                v = re.search(r"VGPRs: (\d+)", blk).group(1)
                sp = re.search(r"VGPRs Spill: (\d+)", blk).group(1)
                info.append("vgpr=%s spill=%s" % (v, sp))
        print(spec, info)

def run(workloads):
    for f in sorted(os.listdir(VDIR)):
        if not f.endswith(".so"): continue
        for wl in workloads:
            env = dict(os.environ, WS_STEREO_LIB=os.path.join(VDIR, f))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "20",
                                "--warmup", "3", "--no-cpu-baseline", "--no-extras", "--check"], env=env, capture_output=True, text=True)
            try:
                j = json.loads(r.stdout.strip().split("\n")[-1])
                print("%-36s %-8s %10.0f Mdisp/s in flight  %10.0f alone  step %.3f ms  kernel %.3f ms  check=%s" % (
                    f, wl, j["value"], j.get("value_single_pair") or 0, j["ms_per_step"], j["roofline"]["kernel_ms"], j.get("check_rows_equal")), flush=True)
            except Exception as e:
                print(f, wl, "FAILED", r.stdout[-300:], r.stderr[-600:], flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "build": build(sys.argv[2:])
    else: run(sys.argv[2:])
