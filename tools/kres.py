#!/usr/bin/env python3
"""Register / spill table of every kernel of a translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py ws_march.hip [grep-pattern]   (CPU only: hipcc cross-compiles gfx950)"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from stereo_reconstruction_amd import build as b

def main():
    name = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    cmd = [b.hipcc()] + b.FLAGS + b.EXTRA.get(name, []) + ["-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c",
           os.path.join(b.CSRC, name), "-o", "/tmp/kres.%d.o" % os.getpid()] + sys.argv[3:]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    os.path.exists("/tmp/kres.%d.o" % os.getpid()) and os.remove("/tmp/kres.%d.o" % os.getpid())
    cur = None
    rows = []
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()}
            rows.append(cur)
            continue
        for key, rx in (("vgpr", r" VGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(rx, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
        if "error" in line:
            print(line)
    for r in rows:
        if pat and not pat.search(r["name"]):
            continue
        n = re.sub(r"wsamd::|\(wsamd::MarchArgs\)|void ", "", r["name"])
        print("%-64s vgpr %3d sgpr %3d  sgpr-spill %3d vgpr-spill %3d scratch %4d occ %d" % (
            n[:64], r.get("vgpr", -1), r.get("sgpr", -1), r.get("sspill", -1), r.get("vspill", -1), r.get("scratch", -1), r.get("occ", -1)))

if __name__ == "__main__":
    main()
