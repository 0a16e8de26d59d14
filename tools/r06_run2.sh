#!/bin/bash
# Round-6 GPU pass 2: full GPU tests (no -x), cfg4 matcher stress (the new Hungarian kernel), host-side cProfile at cfg2 and at the launch-script configuration.
TAG=${1:-r06b}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $OUT/tests_gpu.log 2>&1
grep -v "^HIP\|^ROCm\|^Hostname\|^Librccl\|^$" $OUT/tests_gpu.log | tail -25
grep "attention backward vs fp64\|flash forward vs fp64" $OUT/tests_gpu.log > $OUT/attn_fp64.txt
python tools/bench_cfg4.py > $OUT/cfg4.json 2> $OUT/cfg4.err; tail -5 $OUT/cfg4.json
python tools/debug/host_profile.py > $OUT/host_profile_cfg2.txt 2>&1
head -60 $OUT/host_profile_cfg2.txt | tail -45
