#!/bin/bash
# usage: tools/pool_stress.sh  -- the copy pool of ws_capi.cpp under ThreadSanitizer (CPU only)
R=$(cd $(dirname $0)/.. && pwd)
python3 - <<PY
s = open("$R/stereo_reconstruction_amd/csrc/ws_capi.cpp").read()
body = s[s.index("enum CopyKind {"):s.index("void stage_copy(uint8_t *dst")]
open("/tmp/pool_stress_gen.cpp", "w").write(open("$R/tools/pool_stress.cpp").read().replace("//@POOL@", body))
PY
g++ -O2 -std=c++17 -pthread -mavx2 -fsanitize=thread /tmp/pool_stress_gen.cpp -o /tmp/pool_stress && WS_COPY_THREADS=${WS_COPY_THREADS:-6} /tmp/pool_stress
