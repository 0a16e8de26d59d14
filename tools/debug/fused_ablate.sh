#!/bin/bash
# isolated timings of backward passes 1 / 2 (spe_talking_fused modes 2 / 3) for every ablation build under build_ab/ (tools/ab.py)
for so in build_ab/*.so; do
  n=$(basename $so .so)
  r=$(SPE_HIP_LIB=$so timeout 120 python tools/time_fused.py 2>/dev/null | grep -E "^mode [23]" | tr '\n' ' ')
  echo "$n: $r"
done
