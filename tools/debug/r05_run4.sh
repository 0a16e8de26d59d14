cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
timeout 600 python tools/debug/bwdq_check.py > gpurun_out/r05d/bwdq_check.txt 2>&1; echo "bwdq_check rc=$?"; grep -v "amdgpu.ids" gpurun_out/r05d/bwdq_check.txt | cut -c1-230
for m in 0 1 2 0 2; do
SPE_BWDQ=$m timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05d/bench_$m.json 2>/dev/null; python - <<PY
import json
r=json.loads(open("gpurun_out/r05d/bench_$m.json").read().strip().splitlines()[-1]); print("BWDQ=$m", round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms")
PY
done
SPE_BWDQ=2 timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05d/tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05d/tests_gpu.log
