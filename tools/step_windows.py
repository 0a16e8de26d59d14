"""Per training step of a rocprofv3 --kernel-trace CSV (steps delimited by the last adamw_flat_kernel of each optimiser step):
number of kernels, sum of kernel durations, span, idle split into short gaps (< 20 us: back-to-back dispatch) and long ones.
usage: python tools/step_windows.py <kernel_trace.csv>"""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith("adamw_flat_kernel") and (i + 1 == len(rows) or not rows[i + 1][2].startswith("adamw_flat_kernel"))]
for a, b in zip(ends[:-1], ends[1:]):
    w = rows[a + 1:b + 1]
    busy = 0; short = 0; longg = 0; nshort = 0; be = w[0][1]; busy = w[0][1] - w[0][0]
    for s, e, _ in w[1:]:
        if s > be:
            g = s - be
            if g < 20000: short += g; nshort += 1
            else: longg += g
        busy += max(0, e - max(s, be)); be = max(be, e)
    print("step: %d kernels, busy %.2f ms, span %.2f ms, short gaps %d = %.2f ms (mean %.1f us), long gaps %.2f ms" %
          (len(w), busy / 1e6, (w[-1][1] - w[0][0]) / 1e6, nshort, short / 1e6, short / 1e3 / max(nshort, 1), longg / 1e6))
