"""Instruction histogram of the basic blocks of one kernel in a hipcc -S listing.
    python tools/debug/isa_hist.py fused.s <first line> <last line> [min block size]"""
import re, sys, collections
f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
mn = int(sys.argv[4]) if len(sys.argv) > 4 else 80
lines = open(f).read().split("\n")[a:b]
blocks, cur, name = [], [], "entry"
for l in lines:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\S+):", s)
        if m:
            blocks.append((name, cur)); cur, name = [], m.group(1)
        continue
    cur.append(s.split()[0])
blocks.append((name, cur))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return op
    if op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rcp"): return "trans"
    if op.startswith("v_"): return op
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "ds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_"): return op.split("_")[0] + "_" + op.split("_")[1]
    return op
for name, ops in blocks:
    if len(ops) < mn: continue
    h = collections.Counter(cls(o) for o in ops)
    print(f"== {name}: {len(ops)} instrs")
    print("   " + ", ".join(f"{k}:{v}" for k, v in h.most_common(30)))
