import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "contract" if "attn_contract_kernel<3, false" in n else "contractT" if "attn_contract_kernel<3, true" in n else "aten_sum" if "reduce_kernel" in n else None
        if k is None: continue
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:10s} {c:45s} {v / n:16.1f}  (n={n})")
