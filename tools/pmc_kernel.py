"""Per-kernel averages of the counters in a rocprofv3 --pmc output directory:
    python tools/pmc_kernel.py DIR [kernel-name substring]"""
import collections
import csv
import glob
import sys

if __name__ == "__main__":
    d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if pat in k:
                acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} n={len(v):3d}  avg {sum(v) / len(v):16.1f}")
