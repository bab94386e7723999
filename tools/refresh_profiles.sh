#!/bin/bash
# usage (GPU box): tools/refresh_profiles.sh <tag>  -- bench line + rocprofv3 kernel stats for configs 2 and 3
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
python3 $R/bench.py > $O/config2_bench.json 2> $O/config2_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -o run -- python3 $R/bench.py --workload config2 --steps 20 --no-cpu-baseline > $O/c2.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o run -- python3 $R/bench.py --workload config3 --steps 20 --no-cpu-baseline > $O/c3.log 2>&1 || exit 1
cp $O/c2/run_kernel_stats.csv $O/config2_kernel_stats.csv
grep '^{' $O/c2.log | tail -1 > $O/config2_bench_profiled.json
cp $O/c3/run_kernel_stats.csv $O/config3_kernel_stats.csv
python3 $R/tools/time_calls.py > $O/time_calls.txt 2>&1
tail -1 $O/config2_bench.json | cut -c1-300
