#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R/tools/ubench && hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o /tmp/pcie_probe pcie_probe.hip -lpthread && /tmp/pcie_probe | tee $R/gpurun_out/$1
