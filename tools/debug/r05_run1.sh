cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 600 python tools/debug/bwdq_check.py > gpurun_out/r05a/bwdq_check.txt 2>&1; echo "bwdq_check rc=$?"
tail -40 gpurun_out/r05a/bwdq_check.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r05a/tests_gpu.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05a/bench_new.json 2> gpurun_out/r05a/bench_new.err; tail -c 600 gpurun_out/r05a/bench_new.json
SPE_BWDQ=0 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r05a/bench_old.json 2>/dev/null; tail -c 300 gpurun_out/r05a/bench_old.json
timeout 600 bash tools/debug/fused_pmc.sh r05a/fused_pmc > gpurun_out/r05a/fused_pmc.txt 2>&1; tail -50 gpurun_out/r05a/fused_pmc.txt
timeout 900 bash tools/debug/fused_ablate.sh > gpurun_out/r05a/fused_ablate.txt 2>&1; cat gpurun_out/r05a/fused_ablate.txt
