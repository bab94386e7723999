"""Summarise rocprofv3 --pmc passes (tools/pmc.sh) into profiles/<round>/: the per-kernel counter means as text
and traffic_<workload>.json, the file bench.py reads `roofline.traffic` / `valu_issue` from.

    python tools/pmc_to_traffic.py gpurun_out/<pmc dir> <workload> profiles/r02 [valu_rate.txt]

HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE
counts 64 B per 128-B request for wide (16 B per lane) coalesced reads, which is what the marching kernel's
LDS-DMA row copies are (MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is taken as it is.
The VALU issue ceiling is read from the raw output of tools/ubench/valu_rate.hip when it is given: the best
sustained rate of the kernel's own instruction mix.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main():
    pmc_dir, workload, out_dir = sys.argv[1], sys.argv[2], sys.argv[3]
    ubench = sys.argv[4] if len(sys.argv) > 4 else None
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(pmc_dir, "p*", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = []
    for k, d in sorted(agg.items()):
        lines.append(k[:150])
        for c, v in sorted(d.items()):
            lines.append("   %-28s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "pmc_%s.txt" % workload), "w") as f:
        f.write("# rocprofv3 --pmc, one counter group per pass (tools/pmc.sh), bench.py --workload %s; means per launch\n" % workload)
        f.write("\n".join(lines) + "\n")
    march = [k for k in agg if "ws_march_kernel" in k]
    if not march:
        raise SystemExit("no marching kernel in " + pmc_dir)
    k = max(march, key=lambda n: sum(agg[n].get("SQ_INSTS_VALU", [0])))
    # A search wider than one pass launches the kernel once per d-group pass: bench.py times the search, so the
    # per-launch means are scaled to launches per search (counted against the pre-pass, one launch per search).
    pre = [n for n in agg if "ws_prepare_kernel" in n]  # (rounds 1-3: one pre-pass launch per search; round 4: WS_PASSES)
    per_search = 1
    if pre and agg[pre[0]].get("SQ_INSTS_VALU") and agg[k].get("SQ_INSTS_VALU"):
        per_search = max(1, round(len(agg[k]["SQ_INSTS_VALU"]) / len(agg[pre[0]]["SQ_INSTS_VALU"])))
    if os.environ.get("WS_PASSES"):  # round 4: no pre-pass to count searches by -- the plan's d-group passes, given by the caller
        per_search = int(os.environ["WS_PASSES"])
    extensive = {"FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAVES"}
    m = {c: sum(v) / len(v) * (per_search if c in extensive else 1) for c, v in agg[k].items()}
    t = re.search(r"ws_march_kernel<(\d+), (\d+), (\d+), (\d+), (true|false)", k)
    # (the name the library reports: ",nd4" marks the instantiation with 4 disparities per thread)
    # (",halo": the packed SAD kernel with the halo exchange, the template's last argument)
    halo = re.search(r"ws_march_kernel<[^>]*, (true|false), (true|false)>", k)
    name = "ws_march_kernel<%s,%sx%s%s%s>" % ("ssd" if t.group(5) == "true" else "sad", t.group(3), t.group(4),
                                              ",nd4" if t.group(2) == "4" else "",
                                              ",halo" if halo and halo.group(2) == "true" else "")
    peak, peak_src = None, None
    if ubench and os.path.exists(ubench):
        rates = [float(x) for x in re.findall(r"march-mix \w+\s+waves/SIMD=\d+ :.*?([\d.]+) Tlane-op/s", open(ubench).read())]
        if rates:
            peak, peak_src = max(rates) * 1e12, "best sustained rate of the marching step's own instruction mix in %s" % ubench
    sha_file = os.path.join(pmc_dir, "kernel_sources.sha1")  # written on the GPU box by tools/pmc.sh
    sha = open(sha_file).read().strip() if os.path.exists(sha_file) else None
    out = {
        "workload": workload, "kernel": name, "kernel_sources_sha1": sha, "kernel_symbol": k[:120], "launches_per_search": per_search,
        "source": "%s (rocprofv3 --pmc, FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes; tools/pmc_to_traffic.py)" % pmc_dir,
        "fetch_size_kb": m.get("FETCH_SIZE"), "write_size_kb": m.get("WRITE_SIZE"),
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) coalesced reads -> doubled "
                      "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is",
        "hbm_bytes_per_launch": (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in m and "WRITE_SIZE" in m else None,
        "sq_insts_valu": m.get("SQ_INSTS_VALU"), "sq_waves": m.get("SQ_WAVES"),
        "valu_busy_per_wave": (m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in m and "SQ_ACTIVE_INST_VALU" in m else None,
        "wait_any_per_wave": (m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]) if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m else None,
        "lds_bank_conflict_frac": (m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]) if m.get("SQ_LDS_IDX_ACTIVE") else None,
        "valu_note": "SQ_INSTS_VALU = wave instructions per search (all its d-group passes); x64 lanes = lane-ops",
        "valu_issue_peak_lane_ops_per_s": peak, "valu_peak_source": peak_src,
    }
    with open(os.path.join(out_dir, "traffic_%s.json" % workload), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
