#!/bin/bash
# usage (GPU box): tools/ab_variants.sh <out-file-under-gpurun_out> "<case> <case> ..."
#   every library under gpurun_variants/ (tools/variants.py build ...) and the tree's own, each with the marching
#   kernel's 8- and 4-disparities-per-thread instantiation forced (WS_MARCH_ND), through tools/two_in_flight.py
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1; shift
cases="$@"
: > $out
for lib in $R/stereo_reconstruction_amd/libws_stereo.so $(ls $R/gpurun_variants/*.so 2>/dev/null); do
  for nd in 8 4; do
    echo "== $(basename $lib) WS_MARCH_ND=$nd" >> $out
    WS_STEREO_LIB=$lib WS_MARCH_ND=$nd python $R/tools/two_in_flight.py $cases 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
