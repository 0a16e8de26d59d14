#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "0 d2" "1 d2" "1 d4" "0 d2" "1 d2" "1 d4"; do
  set -- $cfg
  echo -n "ring=$1 lib=$2 : "
  SPE_CONTRACT_LDS=$1 SPE_HIP_LIB=build_ab/$2.so python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],2), round(r['ms_per_step'],2), round(r['roofline']['hbm_kernel']['avg_ms'],4))"
done
