#!/bin/bash
# rocprofv3 kernel statistics of the default bench on the GPU box: tools/prof_stats.sh <tag> [bench args...] -> gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-prof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/prof_$TAG.log 2>&1
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_kernel_stats.csv
find $OUT/prof_$TAG -name "*kernel_trace.csv" -delete
python - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step %.2f ms (3 timed + 1 warm-up + 1 counted step profiled)" % (tot / 5e6))
for r in rows[:45]:
    print("%8.3f ms/step %8.1f us x %7.1f  %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 5e6, float(r["AverageNs"]) / 1e3, float(r["Calls"]) / 5, float(r["Percentage"]), r["Name"][:110]))
PY
