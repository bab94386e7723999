#!/bin/bash
# usage (GPU box): tools/ubench/run.sh <out-file-under-gpurun_out>  -- raw VALU issue rates (copied to profiles/rNN/)
R=$GRAFT_REPO_ROOT
cd $R/tools/ubench && hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/valu_rate valu_rate.hip && /tmp/valu_rate | tee $R/gpurun_out/$1
