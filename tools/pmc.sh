#!/bin/bash
# usage (on the GPU box): tools/pmc.sh <outdir-name> [bench args...]   -- one rocprofv3 --pmc pass per counter group
set -e
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
(cd $GRAFT_REPO_ROOT && python3 -c "import bench; print(bench.kernel_sources_sha1())") > $out/kernel_sources.sha1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --pairs-per-step 8 --no-cpu-baseline --no-extras "$@" > $out/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%d mean=%.4g" % (c, len(v), sum(v)/len(v)))
PY
