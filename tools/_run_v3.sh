rm -f gpurun_out/r4_qt5.txt
for ag in 0 1 4 5; do for fw in 0 1 2 5; do
  WS_STAGE_AGAP=$ag WS_STAGE_WAVE=2 WS_FLUSH_WAVE=$fw timeout -k 5 120 python tools/quick_time.py 2,5 20 >> gpurun_out/r4_qt5.txt 2>&1 || exit 1
done; done
for sw in 0 1 2 3; do for fw in 0 1 2 3; do
  WS_STAGE_WAVE=$sw WS_FLUSH_WAVE=$fw timeout -k 5 120 python tools/quick_time.py 3 20 >> gpurun_out/r4_qt5.txt 2>&1 || exit 1
done; done
grep config gpurun_out/r4_qt5.txt | sort -k7 -n | awk '{print $2, $7, $14, $15, $16, $17}' 
