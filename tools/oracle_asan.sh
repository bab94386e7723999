#!/bin/bash
# usage (build container, CPU): tools/oracle_asan.sh  -- the oracle's tests under AddressSanitizer + UBSan
# (oracle/Makefile: asan).  The sanitized library is loaded instead of libws_oracle.so (WS_ORACLE_LIB).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $R/oracle asan
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export WS_ORACLE_LIB=$R/oracle/libws_oracle_asan.so
cd $R
python3 - <<'PY'
from oracle import oracle
oracle.lib()
print("loaded:", [l.split()[-1] for l in open("/proc/self/maps") if "libws_oracle" in l][0])
PY
python3 -m pytest tests/test_oracle_construction.py tests/test_oracle_golden.py -q 2>&1 | tail -3
