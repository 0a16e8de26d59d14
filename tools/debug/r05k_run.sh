mkdir -p gpurun_out/r05k1
timeout 900 python -m pytest tests/test_round5_gpu.py -x -q 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" | tail -15 > gpurun_out/r05k1/tests_r5.txt
timeout 300 python tools/debug/bwdk_time.py > gpurun_out/r05k1/bwdk_time.txt 2>&1
SPE_BWDQ=3 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05k1/bench_3.json 2>gpurun_out/r05k1/bench_3.err
SPE_BWDQ=2 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05k1/bench_2.json 2>/dev/null
SPE_BWDQ=3 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05k1/bench_3b.json 2>/dev/null
SPE_BWDQ=2 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05k1/bench_2b.json 2>/dev/null
cat gpurun_out/r05k1/tests_r5.txt | tail -6; cat gpurun_out/r05k1/bwdk_time.txt | tail -5
for f in gpurun_out/r05k1/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms")
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
