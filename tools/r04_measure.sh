#!/bin/bash
# Round-4 measurement pass on the GPU box (run through gpurun from the repo root): GPU tests, bench lines, rocprofv3 kernel statistics,
# the two PMC passes the roofline `traffic` fields are computed from, the flash-kernel SQ counters and the decoder window.
# Everything lands under gpurun_out/$1/; the summaries are copied to profiles/ by hand.
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $OUT/tests_gpu.log 2>&1
grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" $OUT/tests_gpu.log | tail -3
python tools/parity_summary.py "round 4 ($TAG)" > $OUT/parity_summary.txt 2>&1; cp profiles/parity_r04.json $OUT/
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --precision bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2>/dev/null
python bench.py --enc-layers 3 --no-cpu-baseline > $OUT/bench_enc3.json 2>/dev/null
SPE_FLASH=0 SPE_MEMKV=0 python bench.py --no-cpu-baseline > $OUT/bench_r03_paths.json 2>/dev/null
python bench.py --backbone TSCAM_cait_S36 --layer-to-det 35 --height 1000 --width 1600 --batch 1 --no-cpu-baseline > $OUT/bench_cfg5.json 2>/dev/null
python bench.py --enc-layers 3 --queries 300 --drop-path 0.2 --attn-drop 0.05 --backbone-drop 0.07 --no-cpu-baseline > $OUT/bench_script_rates_s24.json 2>$OUT/bench_script_rates_s24.err
python tools/debug/cagemm_check.py 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" > $OUT/cagemm.txt
python tools/debug/flash_check.py 2>&1 | grep -v "^HIP\|^ROCm\|amdgpu.ids" | tail -12 > $OUT/flash_check.txt
python tools/bench_cfg4.py > /dev/null 2>&1; cp gpurun_out/r04_cfg4.json $OUT/ 2>/dev/null
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_to_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/roofline_inputs.json "round 4 PMC passes ($TAG): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py --steps 2" > /dev/null 2>&1
python tools/aten_report.py --top 60 2>&1 | grep -v "^\[W\|Warn\|_warn" > $OUT/aten_report.txt
cp $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
bash tools/debug/decoder_window.sh > $OUT/decoder_window.txt 2>&1
bash tools/debug/flash_pmc.sh $TAG/flash_pmc > $OUT/flash_pmc.txt 2>&1
python tools/host_time.py 2>&1 | tail -2 > $OUT/host_time.txt
# keep only the summaries (the traces are large)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms")
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
