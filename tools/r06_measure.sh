#!/bin/bash
# Round-6 measurement pass on the GPU box (run through gpurun from the repo root): GPU tests, bench lines (default, 2-core budget, secondary
# configurations, 2-rank gloo), rocprofv3 kernel statistics, the two PMC passes behind `roofline.traffic`, the step's windows, ATen report, host enqueue
# time, the matcher stress (cfg4), the attention kernels in isolation.  Everything lands under gpurun_out/$1/; the summaries are copied to profiles/ by hand.
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $OUT/tests_gpu.log 2>&1
grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" $OUT/tests_gpu.log | tail -3
python -m pytest tests/test_round6_gpu.py tests/test_round2_gpu.py -m gpu -q -s -k "fp64" 2>&1 | grep "vs fp64" > $OUT/attn_fp64.txt
python tools/parity_summary.py "round 6 ($TAG)" > $OUT/parity_summary.txt 2>&1; cp profiles/parity_r06.json $OUT/ 2>/dev/null
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
taskset -c 0,1 python bench.py --no-cpu-baseline > $OUT/bench_2cores.json 2>/dev/null
python bench.py --no-cpu-baseline > $OUT/bench_unpinned.json 2>/dev/null
# two ranks on the one GPU over gloo (the data-parallel path end to end: bucketed all-reduce beside the backward, num_boxes all-reduce), fp32 and bf16 wire
SPE_BENCH_BACKEND=gloo SPE_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_gloo2.json 2>$OUT/bench_gloo2.err
SPE_BENCH_BACKEND=gloo SPE_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --wire bf16 > $OUT/bench_gloo2_bf16wire.json 2>$OUT/bench_gloo2_bf16wire.err
python bench.py --enc-layers 3 --no-cpu-baseline > $OUT/bench_enc3.json 2>/dev/null
python bench.py --backbone TSCAM_cait_S36 --layer-to-det 35 --height 1000 --width 1600 --batch 1 --no-cpu-baseline > $OUT/bench_cfg5.json 2>/dev/null
python bench.py --enc-layers 3 --queries 300 --drop-path 0.2 --attn-drop 0.05 --backbone-drop 0.07 --no-cpu-baseline > $OUT/bench_script_rates_s24.json 2>$OUT/bench_script_rates_s24.err
python bench.py --backbone TSCAM_cait_XXS36_Two_Branch --layer-to-det 24 --enc-layers 3 --queries 300 --height 512 --width 512 --batch 1 --drop-path 0.2 --attn-drop 0.05 --backbone-drop 0.07 --no-cpu-baseline > $OUT/bench_script_voc.json 2>$OUT/bench_script_voc.err
python tools/bench_cfg4.py > $OUT/cfg4.json 2> $OUT/cfg4.err
python tools/debug/attn_time.py > $OUT/attn_time.txt 2>&1
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_to_json.py $OUT/pmc_fetch $OUT/pmc_write $OUT/roofline_inputs.json "round 6 PMC passes ($TAG): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py --steps 2" > /dev/null 2>&1
python tools/aten_report.py --top 60 2>&1 | grep -v "^\[W\|Warn\|_warn" > $OUT/aten_report.txt
cp $(find $OUT/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
f=$(find $OUT/prof_stats -name "*kernel_trace.csv" | head -1)
python tools/debug/decoder_window.py $f > $OUT/decoder_window.txt 2>&1
python tools/step_windows.py $f >> $OUT/decoder_window.txt 2>&1
python tools/host_time.py 2>&1 | tail -2 > $OUT/host_time.txt
taskset -c 0,1 python tools/host_time.py 2>&1 | tail -2 > $OUT/host_time_2cores.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms")
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
cat $OUT/host_time.txt
