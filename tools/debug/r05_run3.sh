cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 600 python tools/debug/bwdq_check.py > gpurun_out/r05c/bwdq_check.txt 2>&1; echo "bwdq_check rc=$?"; grep -v "amdgpu.ids" gpurun_out/r05c/bwdq_check.txt | cut -c1-260
timeout 900 bash tools/debug/bwdq_ablate.sh > gpurun_out/r05c/bwdq_ablate.txt 2>&1; sed 's/D=.*dbw=[^ ]* //' gpurun_out/r05c/bwdq_ablate.txt
timeout 600 bash tools/debug/fused_pmc.sh r05c/bwdq_pmc tools/debug/bwdq_only.py > gpurun_out/r05c/bwdq_pmc.txt 2>&1; grep "bwdq_kernel" gpurun_out/r05c/bwdq_pmc.txt | grep "STATS\|WAIT\|ACTIVE_INST_ANY\|WAVE_CYC\|MFMA_BUSY\|ACTIVE_INST_VALU"
