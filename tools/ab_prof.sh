#!/bin/bash
# usage (GPU box): tools/ab_prof.sh <tag> <one_call.py args...>  -- kernel stats of one configuration for the tree's library
# and every variant under gpurun_variants/ (200 calls each)
tag=$1; shift
R=$GRAFT_REPO_ROOT
export WS_CALLS=200
for lib in $R/stereo_reconstruction_amd/libws_stereo.so $(ls $R/gpurun_variants/*.so 2>/dev/null); do
  export WS_STEREO_LIB=$lib
  echo "== $(basename $lib)"
  $R/tools/call_prof.sh ${tag}_$(basename $lib .so) "$@" | grep -v "^$"
done
