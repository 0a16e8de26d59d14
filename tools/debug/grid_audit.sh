#!/bin/bash
# Workgroup counts of the kernels of one bench step (rocprofv3 kernel trace): which launches under-fill the chip or spill a nearly
# empty extra round over the resident workgroup slots.  tools/debug/grid_audit.sh [bench args]
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/ga
rocprofv3 --kernel-trace --output-format csv -d /tmp/ga -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /tmp/ga.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/ga/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
t0 = [int(r["Start_Timestamp"]) for r in rows]
# keep the last step: after the last adamw_flat-but-one
idx = [i for i, r in enumerate(rows) if "adamw_flat" in r["Kernel_Name"]]
ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] != i + 1]
lo = ends[-2] + 1 if len(ends) >= 2 else 0
hi = ends[-1] + 1
d = collections.defaultdict(list)
for r in rows[lo:hi]:
    wg = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    grid = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    d[(r["Kernel_Name"][:70], grid // wg, wg, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in d.values())
print("step: %d launches, %.2f ms" % (hi - lo, tot / 1e3))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:60]:
    print("%8.1f us total %4d x %7.1f us  wgs %6d x %4d thr  lds %6s vgpr %4s  %s" % (sum(v), len(v), sum(v) / len(v), k[1], k[2], k[3], k[4], k[0]))
PY
