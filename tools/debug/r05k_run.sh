# round-5 development pass on the GPU box: bash tools/debug/r05k_run.sh <tag> [pytest -k filter]
TAG=${1:-r05k}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q ${2:+-k "$2"} 2>&1 | grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" | tail -40 > $OUT/tests.txt
for m in 1 0 1 0; do
  python - $m > $OUT/bench_mlp$m.$RANDOM.json 2>/dev/null <<'PY'
import sys, runpy
from spe_amd import kernels as K
K.MLP_F16 = sys.argv[1] == "1"
sys.argv = ["bench.py", "--no-cpu-baseline"]
runpy.run_path("bench.py", run_name="__main__")
PY
done
bash tools/prof_stats.sh $TAG > $OUT/kstats.txt 2>&1
tail -12 $OUT/tests.txt
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms")
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
