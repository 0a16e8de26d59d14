#!/bin/bash
# rocprofv3 kernel durations of tools/bench_small.py (per-shape small-row Linear): tools/debug/small_prof.sh
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python $GRAFT_REPO_ROOT/tools/bench_small.py 2>&1 | grep -E "^R=|SPE_LINEAR"
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/ps/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-80s calls %6s avg %8.1f us min %8.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
t = glob.glob("/tmp/ps/**/*kernel_trace.csv", recursive=True)[0]
import collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(t)):
    if "linear_small" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:40], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    v.sort()
    print(k, "n=%d median %.1f us min %.1f us" % (len(v), v[len(v) // 2], v[0]))
PY
