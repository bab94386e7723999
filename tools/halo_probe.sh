#!/bin/bash
# usage (GPU box): tools/halo_probe.sh -- workgroup sizes for the bench workloads (WS_PLAN_THREADS), with / without the halo-exchange SAD kernel
run() { # workload, env...
  wl=$1; shift
  echo "== $wl $*"
  env "$@" python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras --check 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['value'], j.get('value_single_pair'), j['ms_per_step'], j['roofline']['kernel_ms'], j.get('check_rows_equal'))"
}
run config2 WS_PLAN_THREADS=0
run config2 WS_PLAN_THREADS=256
run config3 WS_PLAN_THREADS=384
run config3 WS_PLAN_THREADS=256 WS_PLAN_SLOTS=2
run config5 WS_PLAN_THREADS=0
run config5 WS_PLAN_THREADS=256
