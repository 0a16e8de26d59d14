mkdir -p gpurun_out/r05k6
python -m pytest tests/test_round5_gpu.py tests/test_kernels_gpu.py -q -k "rides_on or layer_norm or layernorm" 2>&1 | tail -2
for m in 1 0 1 0; do
  python - $m > gpurun_out/r05k6/bench_lnls$m.$RANDOM.json 2>/dev/null <<'PY'
import sys, runpy
from spe_amd import kernels as K
K.LN_LS_FUSE = sys.argv[1] == "1"
sys.argv = ["bench.py", "--no-cpu-baseline"]
runpy.run_path("bench.py", run_name="__main__")
PY
done
bash tools/prof_stats.sh r05k6 > gpurun_out/r05k6/kstats.txt 2>&1
for f in gpurun_out/r05k6/bench_*.json; do python - "$f" <<'PY'
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms")
PY
done
grep "ln_bwd\|lsres" gpurun_out/r05k6/kstats.txt
