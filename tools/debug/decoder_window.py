"""Kernels of ONE step by window: backbone forward (up to the last flash forward + merge), decoder + criteria + class attention both ways (up to the
first key-major attention backward kernel), backbone backward (up to the last dK contraction), tail (patch embed backward, all-reduce, optimiser).  usage: decoder_window.py <kernel_trace.csv>"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith("adamw_flat_kernel") and (i + 1 == len(rows) or not rows[i + 1][2].startswith("adamw_flat_kernel"))]
a, b = ends[-2] + 1, ends[-1] + 1
w = rows[a:b]
def short(n):
    n = n.replace("void ", "")
    return n[:70]
is_flash_fwd = lambda n: "talking_flash_fwd_kernel" in n
is_stats = lambda n: "talking_stats_kernel" in n
is_bwdk = lambda n: "talking_bwdk_kernel" in n           # first kernel of a block's attention backward
is_bwdq = lambda n: "talking_bwdq_kernel" in n
m1 = max(i for i, r in enumerate(w) if is_flash_fwd(r[2]))
fwd_end = max(i for i, r in enumerate(w) if "flash_merge_kernel" in r[2] and i < m1 + 3)
m2 = min(i for i, r in enumerate(w) if is_bwdk(r[2]))
# the last block's backward ends with its dK contraction (attn_contract_kernel<3, true, ...>) a few launches behind its query-major kernel
mq = max(i for i, r in enumerate(w) if is_bwdq(r[2]))
m3 = max([i for i, r in enumerate(w) if "attn_contract_kernel" in r[2] and i < mq + 8] + [mq])
for title, lo, hi in (("backbone forward", 0, fwd_end + 1), ("decoder + criteria + class attention (fwd and bwd)", fwd_end + 1, m2), ("backbone backward", m2, m3 + 1), ("tail: stem backward, optimiser", m3 + 1, len(w))):
    seg = w[lo:hi]
    tot = sum(e - s for s, e, _ in seg) / 1e3
    span = (seg[-1][1] - seg[0][0]) / 1e6 if seg else 0.0
    print("== %s: %d kernels, %.2f ms busy, %.2f ms span" % (title, len(seg), tot / 1e3, span))
    if "decoder" in title or "tail" in title:
        c = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n in seg:
            c[short(n)][0] += 1; c[short(n)][1] += (e - s) / 1e3
        for n, (k, t) in sorted(c.items(), key=lambda x: -x[1][1])[:28]:
            print("   %7.1f us %4d  %s" % (t, k, n))
# ---- the launch sequence of ONE backbone block, forward (between two statistics passes) and backward (between two key-major kernels)
def seq(title, idx):
    if len(idx) < 14:
        return
    lo, hi = idx[12], idx[13]
    print("== %s: one block = %d launches, %.1f us busy, %.1f us span" % (title, hi - lo, sum(e - s for s, e, _ in w[lo:hi]) / 1e3, (w[hi][0] - w[lo][0]) / 1e3))
    for s, e, n in w[lo:hi]:
        print("   %7.1f us  %s" % ((e - s) / 1e3, short(n)[:100]))
seq("backbone forward", [i for i, r in enumerate(w) if is_stats(r[2])])
seq("backbone backward", [i for i, r in enumerate(w) if is_bwdk(r[2])])
