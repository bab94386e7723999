#!/bin/bash
# usage (GPU box): tools/ubench/run_hostreg.sh <out-file-under-gpurun_out> [scenario ...]
R=$GRAFT_REPO_ROOT
out=$1; shift
cd $R/tools/ubench && hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/hostreg_probe hostreg_probe.hip && timeout -k 10 400 /tmp/hostreg_probe "$@" > $R/gpurun_out/$out 2>&1
echo "probe exit $?" >> $R/gpurun_out/$out
# one register + copy + unregister with the runtime's own log
# (AMD_LOG_LEVEL=4 in front of the probe shows the runtime's own account: profiles/r03/hostreg_probe_amdlog.txt)
tail -c 3000 $R/gpurun_out/$out
