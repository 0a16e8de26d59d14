# Which launches surround the runtime's copy kernels (__amd_rocclr_copyBuffer) in one training step: (previous kernel, next kernel) pairs by count.
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/cpn
rocprofv3 --kernel-trace --output-format csv -d /tmp/cpn -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/cpn.log 2>&1
f=$(find /tmp/cpn -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = sorted((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
names = [n for _, n in rows]
ends = [i for i, n in enumerate(names) if n.startswith("adamw_flat_kernel") and (i + 1 == len(names) or not names[i + 1].startswith("adamw_flat_kernel"))]
a, b = ends[-2] + 1, ends[-1] + 1
w = names[a:b]
short = lambda n: n.replace("void ", "")[:60]
c = collections.Counter()
for i, n in enumerate(w):
    if "copyBuffer" in n:
        p = next((w[j] for j in range(i - 1, -1, -1) if "copyBuffer" not in w[j]), "-")
        q = next((w[j] for j in range(i + 1, len(w)) if "copyBuffer" not in w[j]), "-")
        c[(short(p), short(q))] += 1
print("copy kernels in the step:", sum(c.values()))
for (p, q), k in c.most_common(30):
    print("%4d  after %-60s before %s" % (k, p, q))
PY
