#!/bin/bash
# usage (GPU box): tools/halo_probe.sh -- config 3 with the halo-exchange SAD kernel per workgroup size (WS_PLAN_THREADS), and without it
run() { # workload, env...
  wl=$1; shift
  echo "== $wl $*"
  env "$@" python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras --check 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = j['roofline']; print(j['value'], j.get('value_single_pair'), j['ms_per_step'], r['kernel_ms'], r.get('kernel'), r.get('threads'), r.get('workgroups'), j.get('check_rows_equal'))"
}
run config3 WS_PLAN_THREADS=0
run config3 WS_PLAN_THREADS=256
run config3 WS_PLAN_THREADS=512
run config3 WS_MARCH_HALO=0
