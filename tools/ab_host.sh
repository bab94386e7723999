# A/B of host-side variants of ws_search_host inside ONE gpurun session (boxes differ): 3 rounds x 30 calls each
for i in 1 2 3; do
for kb in 0 512 1024 2048; do
 echo "chunk_kb=$kb"; WS_HOST_TRACE=0 REPS=30 BANDS=-1,0,4 WS_UP_CHUNK_KB=$kb timeout -k 5 120 python tools/host_trace.py 2>&1 | grep dtype
done; done
