"""Sum rocprofv3 --pmc counter_collection.csv rows per (kernel, counter) for kernels matching a substring.
usage: python tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_pt_filter"
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
tot = defaultdict(float); n = defaultdict(int)
for f in files:
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} {tot[k]:.6g}  ({n[k]} dispatches)")
