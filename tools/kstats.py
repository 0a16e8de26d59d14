"""Per-kernel totals of a rocprofv3 rocpd database: python tools/kstats.py <results.db> [steps] [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print("total kernel time %.2f ms per step (%d steps)" % (tot / 1e3 / steps, steps))
for n, c, d, a, p in rows[:top]:
    print("%8.3f ms/step %7.1f us x %6.1f  %5.1f%%  %s" % (d / 1e3 / steps, a, c / steps, p, n[:90]))
