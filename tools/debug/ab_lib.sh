#!/bin/bash
# same-box A/B of library builds: bash tools/debug/ab_lib.sh <tool.py> <rounds> name1 name2 ... (build_ab/<name>.so from tools/ab.py), interleaved
T=$1; R=$2; shift 2
for r in $(seq $R); do for n in "$@"; do echo "== $n (round $r)"; SPE_HIP_LIB=build_ab/$n.so python $T 2>&1 | grep -v "amdgpu.ids\|^library"; done; done
