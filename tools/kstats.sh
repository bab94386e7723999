#!/bin/bash
# usage (GPU box): tools/kstats.sh <name> [bench args]  -- rocprofv3 kernel stats of a bench.py run
name=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -o run -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $R/gpurun_out/$name.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/$name/run_kernel_stats.csv")):
    print("%-70s calls=%s avg=%.2f us  %s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
tail -1 $R/gpurun_out/$name.log | cut -c1-330
