#!/bin/bash
# per-kernel times of the left view's smoothFactor passes (rocprofv3 kernel stats)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "900 750 17 0 200 0.9" "900 750 17 0 200 1.5" "900 750 7 0 200 0.9" "1500 1000 7 0 256 0.9" "1500 1000 7 0 256 1.5"; do
  set -- $cfg
  tag="w$1_bs$3_s$6"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lsp_$tag -o run -- python3 $R/tools/one_call.py left $1 $2 $3 $4 $5 $6 ssd > $R/gpurun_out/lsp_$tag.log 2>&1
  f=$(find $R/gpurun_out/lsp_$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag" >> $R/gpurun_out/lsp_summary.txt
  cut -d, -f1-4 "$f" | head -8 >> $R/gpurun_out/lsp_summary.txt
done
