#!/bin/bash
# rocprofv3 SQ-counter passes over a launcher script (default tools/debug/fused_only.py: backward passes 1 and 2 of the talking-heads
# attention); prints per-kernel averages of each counter for the attention kernels.  Counters are collected with --kernel-trace only.
#   bash tools/debug/fused_pmc.sh <out-subdir> [launcher.py] [kernel-name substring ...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-fused_pmc}
LAUNCH=${2:-tools/debug/fused_only.py}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/$LAUNCH > $OUT/stats.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python $GRAFT_REPO_ROOT/$LAUNCH > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
want = ("talking_fused_kernel", "talking_flash", "attn_contract", "talking_bwdq_kernel", "talking_bwdk_kernel")
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(w in r["Name"] for w in want):
            print("STATS", r["Name"][:70], r["Calls"], "avg_ns", r["AverageNs"])
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(w in r["Kernel_Name"] for w in want):
            k = (r["Kernel_Name"][:52], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k in sorted(acc):
    print("PMC", k[0], k[1], "%.4g" % (acc[k][0] / acc[k][1]), "n", acc[k][1])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
