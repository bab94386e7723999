#!/bin/bash
# usage (GPU box): tools/nd_rule_check.sh <out-file-under-gpurun_out>  -- the thread-shape rule (ws_march.hip: march_shape)
# against both instantiations forced, at two image sizes other than the one its weight was measured at
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$1
: > $out
cases=""
for size in 900,750 2964,1988; do for bs in 5 9 17; do for D in 64 128 256; do for cost in ssd sad; do cases="$cases $size,$bs,$D,$cost,left"; done; done; done; done
python $R/tools/two_in_flight.py 1500,1000,7,256,ssd,left > /dev/null 2>&1  # (clocks up before the first section)
for nd in 8 4 rule; do
  echo "== WS_MARCH_ND=$nd" >> $out
  if [ $nd = rule ]; then python $R/tools/two_in_flight.py $cases 2>&1 | grep -v amdgpu.ids >> $out
  else WS_MARCH_ND=$nd python $R/tools/two_in_flight.py $cases 2>&1 | grep -v amdgpu.ids >> $out; fi
done
python3 - $out <<'PY'
import re, sys
sec, t = None, {}
for l in open(sys.argv[1]):
    if l.startswith("=="): sec = l.split("=")[-1].strip(); continue
    m = re.match(r"(\S+ \S+ \S+ D=\d+) \S+: ([\d.]+) ms per pair alone, ([\d.]+) with 2", l)
    if m: t.setdefault(m.group(1), {})[sec] = (float(m.group(2)), float(m.group(3)))
worst = 0.0
lines = []
for k, v in t.items():
    if len(v) < 3: continue
    best = min(v["8"][0], v["4"][0])
    loss = v["rule"][0] / best - 1
    worst = max(worst, loss)
    lines.append("%-28s rule %.4f  nd8 %.4f  nd4 %.4f  (alone; in flight %.4f / %.4f / %.4f)  rule vs best alone %+.1f %%" % (k, v["rule"][0], v["8"][0], v["4"][0], v["rule"][1], v["8"][1], v["4"][1], 100 * loss))
open(sys.argv[1], "a").write("\n== summary: the rule's choice against the better of the two forced shapes (ms per pair)\n" + "\n".join(lines) + "\nworst loss alone: %.1f %%\n" % (100 * worst))
print("\n".join(lines[-8:])); print("worst loss alone: %.1f %%" % (100 * worst))
PY
