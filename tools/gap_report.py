"""GPU idle gaps from a rocprofv3 --kernel-trace CSV: total idle time, the largest gaps and the kernels around them.
usage: python tools/gap_report.py <kernel_trace.csv> [min_gap_us] [last_fraction]"""
import csv
import sys


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
    frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.25
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    n0 = int(len(rows) * (1.0 - frac))          # the last steps only (steady state)
    rows = rows[n0:]
    busy_end = rows[0][1]
    gaps = []
    idle = 0
    for i in range(1, len(rows)):
        s, e, name = rows[i]
        if s > busy_end:
            g = (s - busy_end) / 1e3
            idle += g
            if g >= min_gap:
                gaps.append((g, i))
        busy_end = max(busy_end, e)
    span = (rows[-1][1] - rows[0][0]) / 1e3
    print("kernels %d, span %.1f us, idle %.1f us (%.1f %%)" % (len(rows), span, idle, 100 * idle / span))
    small = sum(g for g, _ in gaps)
    print("gaps >= %.0f us: %d, %.1f us" % (min_gap, len(gaps), small))
    gaps.sort(reverse=True)
    for g, i in gaps[:40]:
        print("%9.1f us  after %-60s before %s" % (g, rows[i - 1][2][:60], rows[i][2][:60]))


if __name__ == "__main__":
    main()
