#!/bin/bash
# usage (GPU box): tools/halo_probe.sh -- the bench workloads with / without the halo-exchange kernels (WS_MARCH_HALO=0: none,
# WS_MARCH_HALO_SSD=0: not for SSD) and per workgroup size (WS_PLAN_THREADS)
run() { # workload, env...
  wl=$1; shift
  echo "== $wl $*"
  env "$@" python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-extras --check 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = j['roofline']; print(j['value'], j.get('value_single_pair'), j['ms_per_step'], r['kernel_ms'], r.get('kernel'), j.get('check_rows_equal'))"
}
run config2 WS_X=0
run config2 WS_MARCH_HALO_SSD=0
run config2 WS_PLAN_THREADS=256
run config5 WS_X=0
run config5 WS_MARCH_HALO_SSD=0
run config3 WS_X=0
