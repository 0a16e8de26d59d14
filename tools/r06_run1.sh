#!/bin/bash
# Round-6 GPU pass 1: the GPU tests of the cleaned tree, the default bench line, a kernel trace (per-kernel statistics + the step's windows),
# the attention kernels in isolation for the shipped library and the three ablation builds (what the dS store / the streamed operands cost).
TAG=${1:-r06a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $OUT/tests_gpu.log 2>&1
grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" $OUT/tests_gpu.log | tail -15
grep "attention backward vs fp64\|flash forward vs fp64" $OUT/tests_gpu.log > $OUT/attn_fp64.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
python - <<'PY'
import json,os
f=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/"+os.environ.get("TAG","r06a")+"/bench_default.json"
try:
    r=json.loads(open(f).read().strip().splitlines()[-1]); print("bench", round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms", r["roofline"]["avg_ms"], r["roofline"]["frac"])
except Exception as e: print("bench FAILED", e)
PY
for v in "" sametile nost sametile_nost; do
  if [ -z "$v" ]; then python tools/debug/attn_time.py > $OUT/attn_time_shipped.txt 2>&1; else SPE_HIP_LIB=build_ab/$v.so python tools/debug/attn_time.py > $OUT/attn_time_$v.txt 2>&1; fi
done
tail -n 7 $OUT/attn_time_*.txt
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
f=$(find $OUT/prof_stats -name "*kernel_trace.csv" | head -1)
python tools/debug/decoder_window.py $f > $OUT/decoder_window.txt 2>&1
python tools/step_windows.py $f >> $OUT/decoder_window.txt 2>&1
python tools/host_time.py 2>&1 | tail -2 > $OUT/host_time.txt
cat $OUT/host_time.txt
find $OUT -name "*kernel_trace.csv" -delete
