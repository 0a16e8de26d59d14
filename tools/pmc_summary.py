"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (name shortened), the mean of each counter per launch.

usage: python tools/pmc_summary.py <dir-or-csv> [substring filter]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if filt and filt not in k:
                    continue
                k = k.replace("void ", "")[:70]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            print(f"    {c:32s} {acc[k][c] / cnt[k][c]:16.1f}   (n={cnt[k][c]})")


if __name__ == "__main__":
    main()
