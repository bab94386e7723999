#!/bin/bash
# usage (GPU box): tools/nd_grid.sh <out-file-under-gpurun_out>  -- the table behind march_nd's rule: windows x ranges x costs
# at 1500 x 1000, either instantiation of the marching kernel forced, alone and with two / three pairs in flight
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1
: > $out
for nd in 8 4; do
  echo "== WS_MARCH_ND=$nd" >> $out
  WS_MARCH_ND=$nd python $R/tools/two_in_flight.py --grid 2>&1 | grep -v amdgpu.ids >> $out
done
tail -5 $out
