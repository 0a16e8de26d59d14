#!/bin/bash
# isolated timings of the q-major flash-skeleton backward passes for every variant under build_ab/ (tools/ab.py, ONLY=attn_flash_bwd.hip)
for so in build_ab/*.so; do
  n=$(basename $so .so)
  r=$(SPE_HIP_LIB=$so QUICK=1 timeout 120 python tools/debug/bwdq_check.py 2>/dev/null | grep -E "new pass|old pass|old dQ|finite" | sed 's/ (with.*//' | tr '\n' ' ')
  echo "$n: $r"
done
