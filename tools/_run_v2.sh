tools/gpu_session.sh r4_parity2 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu || exit 1
for sw in -1 4 6; do for fw in -1 4 2; do
  WS_STAGE_WAVE=$sw WS_FLUSH_WAVE=$fw timeout -k 5 120 python tools/quick_time.py 2,3 20 >> gpurun_out/r4_qt2.txt 2>&1 || exit 1
done; done
cat gpurun_out/r4_qt2.txt
