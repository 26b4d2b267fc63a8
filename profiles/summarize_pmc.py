#!/usr/bin/env python3
"""Averages rocprofv3 --pmc CSV output per kernel and counter: summarize_pmc.py DIR"""
import csv, glob, os, re, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", row["Kernel_Name"])
        name = m.group(1) if m else row["Kernel_Name"][:40]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-22s avg %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
