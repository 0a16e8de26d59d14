cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05e
QUICK=1 timeout 300 python tools/debug/bwdq_check.py 2>&1 | grep -v amdgpu.ids | cut -c1-230
for m in 0 2 0 2; do
SPE_BWDQ=$m timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05e/bench_$m.json 2>/dev/null; python - <<PY
import json
r=json.loads(open("gpurun_out/r05e/bench_$m.json").read().strip().splitlines()[-1]); print("BWDQ=$m", round(r["value"],2), "img/s", round(r["ms_per_step"],2), "ms")
PY
done
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05e/tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r05e/tests_gpu.log
timeout 600 bash tools/debug/fused_pmc.sh r05e/fused_pmc > gpurun_out/r05e/fused_pmc.txt 2>&1; grep "STATS\|BANK\|LDS_IDX\|WAVE_CYC" gpurun_out/r05e/fused_pmc.txt
