#!/bin/bash
# usage (GPU box): tools/ab_lib.sh <out-file-under-gpurun_out> <variant .so under gpurun_variants/> "<quick_time cases>" [rounds]
#   the tree's library and ONE variant alternately through tools/quick_time.py in one session (ms per call + map hashes)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1; var=$R/gpurun_variants/$2; cases=$3; rounds=${4:-3}
: > $out
for i in $(seq $rounds); do
  for lib in $R/stereo_reconstruction_amd/libws_stereo.so $var; do
    echo "== $(basename $lib)" >> $out
    WS_STEREO_LIB=$lib python $R/tools/quick_time.py $cases 40 2>&1 | grep -v amdgpu.ids >> $out || exit 1
  done
done
cat $out
