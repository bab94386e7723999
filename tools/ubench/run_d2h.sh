#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R/tools/ubench && hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/d2h_probe d2h_probe.hip && /tmp/d2h_probe | tee $R/gpurun_out/$1
