#!/bin/bash
# usage (GPU box): tools/box_variants.sh  -- the 8-bit box filter's column pass per band width (WS_BOX_COLS), kernel durations
R=$GRAFT_REPO_ROOT
run() { # name, env...
    name=$1; shift
    echo "== $name: $*"
    env "$@" bash $R/tools/consumers_prof.sh $name 2>&1 | grep "outlier\|box_rows\|identical"
}
run boxv_default WS_X=0
for c in 4 8 16; do run boxv_cols$c WS_BOX_COLS=$c; done
