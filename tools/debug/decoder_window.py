"""Kernels of ONE step outside the 24 backbone blocks' attention span: from the last forward PV contraction to the first backward
score pass (decoder + criteria forward and backward, class attention), and after the last backward pass 2 (patch embed backward,
all-reduce, optimiser).  usage: decoder_window.py <kernel_trace.csv>"""
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
def is_flash_fwd(n):          # talking_flash_fwd_kernel<H, DSTEPS, TAIL16, DROP, KV = false>: the forward pass (round 4)
    return "talking_flash_fwd_kernel" in n and n.split("<")[1].split(">")[0].split(",")[4].strip() == "false"
if any(is_flash_fwd(r[2]) for r in w):
    m1 = max(i for i, r in enumerate(w) if is_flash_fwd(r[2]))
    fwd_end = max(i for i, r in enumerate(w) if "flash_merge_kernel" in r[2] and i < m1 + 3)
else:
    m1 = max(i for i, r in enumerate(w) if "talking_fused_kernel" in r[2] and ", 1, " in r[2].split("<")[1][:20])
    fwd_end = max(i for i, r in enumerate(w) if "attn_contract_kernel<3, false" in r[2] and i < m1 + 3)
m2 = min(i for i, r in enumerate(w) if "talking_fused_kernel" in r[2] and r[2].split("<")[1].split(",")[3].strip() == "2")
m3 = max(i for i, r in enumerate(w) if "talking_fused_kernel" in r[2] and r[2].split("<")[1].split(",")[3].strip() == "3")
for title, lo, hi in (("backbone forward", 0, fwd_end + 1), ("decoder + criteria + class attention (fwd and bwd)", fwd_end + 1, m2), ("backbone backward", m2, m3 + 1), ("tail: stem backward, optimiser", m3 + 1, len(w))):
    seg = w[lo:hi]
    tot = sum(e - s for s, e, _ in seg) / 1e3
    print("== %s: %d kernels, %.2f ms busy" % (title, len(seg), tot / 1e3))
    if "decoder" in title or "tail" in title:
        c = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n in seg:
            c[short(n)][0] += 1; c[short(n)][1] += (e - s) / 1e3
        for n, (k, t) in sorted(c.items(), key=lambda x: -x[1][1])[:28]:
            print("   %7.1f us %4d  %s" % (t, k, n))
# ---- the launch sequence of ONE backbone block, forward (between two statistics passes) and backward (between two backward passes 1)
def seq(title, idx):
    if len(idx) < 14:
        return
    lo, hi = idx[12], idx[13]
    print("== %s: one block = %d launches, %.1f us busy, %.1f us span" % (title, hi - lo, sum(e - s for s, e, _ in w[lo:hi]) / 1e3, (w[hi][0] - w[lo][0]) / 1e3))
    for s, e, n in w[lo:hi]:
        print("   %7.1f us  %s" % ((e - s) / 1e3, short(n)[:100]))
mode = lambda r, k: "talking_fused_kernel" in r[2] and r[2].split("<")[1].split(",")[3].strip() == k
seq("backbone forward", [i for i, r in enumerate(w) if mode(r, "0")])
seq("backbone backward", [i for i, r in enumerate(w) if mode(r, "2")])
