for i in 1 2 3; do
for e in 0 1; do for st in 0 1; do
 echo "early=$e stream=$st"; WS_HOST_TRACE=0 REPS=30 BANDS=-1,4 WS_HOST_EARLY=$e WS_COPY_STREAM=$st timeout -k 5 120 python tools/host_trace.py 2>&1 | grep dtype
done; done; done
