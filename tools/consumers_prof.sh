#!/bin/bash
# usage (GPU box): tools/consumers_prof.sh <name>  -- kernel durations of the Reconstruction-side consumers
name=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$name -o run -- python3 $R/tools/time_consumers.py 5 > $R/gpurun_out/$name.log 2>&1
cat $R/gpurun_out/$name.log | grep -v "^W2026\|amdgpu.ids"
python3 $R/tools/consumers_report.py $R/gpurun_out/$name/run_kernel_trace.csv
