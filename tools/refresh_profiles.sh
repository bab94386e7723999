#!/bin/bash
# usage (GPU box): tools/refresh_profiles.sh <tag>  -- bench line + rocprofv3 kernel stats for configs 2 and 3
# (config 2 twice: the default command, two pairs in flight, and --in-flight 1, where the kernels run one after the other)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
python3 $R/bench.py > $O/config2_bench.json 2> $O/config2_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o run -- python3 $R/bench.py --workload config2 --steps 20 --pairs-per-step 16 --no-cpu-baseline --no-extras > $O/c2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2s -o run -- python3 $R/bench.py --workload config2 --steps 20 --pairs-per-step 16 --in-flight 1 --no-cpu-baseline --no-extras > $O/c2s.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o run -- python3 $R/bench.py --workload config3 --steps 20 --pairs-per-step 4 --no-cpu-baseline --no-extras > $O/c3.log 2>&1 || exit 1
cp $O/c2/run_kernel_stats.csv $O/config2_kernel_stats.csv
grep '^{' $O/c2.log | tail -1 > $O/config2_bench_profiled.json
cp $O/c2s/run_kernel_stats.csv $O/config2_inflight1_kernel_stats.csv
grep '^{' $O/c2s.log | tail -1 > $O/config2_inflight1_bench_profiled.json
cp $O/c3/run_kernel_stats.csv $O/config3_kernel_stats.csv
grep '^{' $O/c3.log | tail -1 > $O/config3_bench_profiled.json
python3 $R/tools/time_calls.py > $O/time_calls.txt 2>&1
python3 - <<PY
import csv, json
for t in ("config2", "config2_inflight1", "config3"):
    rows = list(csv.DictReader(open("$O/%s_kernel_stats.csv" % t)))
    b = json.load(open("$O/%s_bench_profiled.json" % t))
    print(t, "value", b["value"], "kernel_ms (bench, alone)", b["roofline"]["kernel_ms"])
    for r in rows[:3]:
        print("   %-70s calls=%s avg=%.2f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 $O/config2_bench.json | cut -c1-300
