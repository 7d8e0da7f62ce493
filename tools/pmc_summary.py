#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc counter_collection.csv files: mean per kernel per counter."""
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in sys.argv[1:]:
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            m = re.search(r"(\w+_kernel)", k)
            short = m.group(1) if m else k[:40]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s n=%-3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
