#!/bin/bash
# rocprofv3 counter passes over tools/debug/attn_time.py; prints per-kernel averages of each counter for the flash kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-flash_pmc}
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/tools/debug/attn_time.py > $OUT/stats.log 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -- python $GRAFT_REPO_ROOT/tools/debug/attn_time.py > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flash" in r["Name"] or "talking" in r["Name"]:
            print("STATS", r["Name"][:60], r["Calls"], "avg_ns", r["AverageNs"])
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "talking_" in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k in sorted(acc):
    print("PMC", k[0], k[1], "%.4g" % (acc[k][0] / acc[k][1]), "n", acc[k][1])
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
