cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
timeout 900 bash tools/debug/bwdq_ablate.sh > gpurun_out/r05b/bwdq_ablate.txt 2>&1; cat gpurun_out/r05b/bwdq_ablate.txt
timeout 600 bash tools/debug/fused_pmc.sh r05b/bwdq_pmc tools/debug/bwdq_only.py > gpurun_out/r05b/bwdq_pmc.txt 2>&1; grep "bwdq_kernel" gpurun_out/r05b/bwdq_pmc.txt
