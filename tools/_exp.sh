for i in 1 2; do
for ag in 0 77; do WS_STAGE_AGAP=$ag timeout -k 5 120 python tools/quick_time.py 2,3,5 20 2>&1 | grep config; done; done
