#!/bin/bash
# usage: ab_env.sh VAR val1 val2 ... : bench A/B of an environment switch (two rounds)
cd $GRAFT_REPO_ROOT
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do
  echo -n "$VAR=$v : "
  env $VAR=$v python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(round(r['value'],2), round(r['ms_per_step'],2))"
done; done
