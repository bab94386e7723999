#!/bin/bash
# usage (GPU box): tools/call_prof.sh <tag> <one_call.py args...>  -- rocprofv3 kernel stats of one configuration
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cp_$tag -o run -- python3 $R/tools/one_call.py "$@" > $R/gpurun_out/cp_$tag.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/cp_$tag/run_kernel_stats.csv")):
    print("%-90s calls=%s avg=%.2f us  %s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
