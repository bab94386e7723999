for e in P; do
  WS_STEREO_LIB=$GRAFT_REPO_ROOT/gpurun_variants/exp$e/libws_stereo.so timeout -k 5 200 python tools/left_smooth_time.py 2>&1 | grep "s=0.9" | sed "s/^/exp$e /"
done
timeout -k 5 200 python tools/left_smooth_time.py 2>&1 | grep "s=0.9" | sed "s/^/tree /"
