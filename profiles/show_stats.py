#!/usr/bin/env python3
"""Prints the per-kernel rows of a rocprofv3 --stats kernel_stats.csv: show_stats.py DIR"""
import csv
import glob
import os
import re
import sys

for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Name"])
        name = m.group(1) if m else r["Name"][:40]
        print("%-34s calls %5s  avg %10.1f ns  min %9s  max %9s" %
              (name, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
