#!/bin/bash
TAG=${1:-r06c}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_round6_gpu.py -m gpu -q -s > $OUT/tests_round6.log 2>&1
grep "vs fp64\|a-resident\|passed\|failed" $OUT/tests_round6.log | tail -40
python -m pytest tests -m gpu -q -x --deselect tests/test_round6_gpu.py > $OUT/tests_gpu.log 2>&1
grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" $OUT/tests_gpu.log | tail -8
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.err
python - <<'PY'
import json,os
f=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r06c/bench_default.json"
try:
    r=json.loads(open(f).read().strip().splitlines()[-1]); print("bench", round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms", r["roofline"]["avg_ms"], r["roofline"]["frac"]); print({k:v for k,v in r["roofline"]["decoder_ca_gemm"].items() if k in ("launches","avg_us","achieved_tflops","mfma_frac","hbm_frac","kernel")})
except Exception as e: print("bench FAILED", e)
PY
python tools/debug/cagemm_check.py > $OUT/cagemm_iso.txt 2>&1; tail -8 $OUT/cagemm_iso.txt
