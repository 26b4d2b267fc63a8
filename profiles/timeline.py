#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints, for the steady part of the run, the mean
duration of every kernel, the span per step and how much of the span had 0 / 1 / 2+ kernels
running: timeline.py DIR STEPS_PER_RUN"""
import csv
import glob
import os
import re
import sys

rows = []
for path in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:30]
        # the y_true instances of the match kernels are kernels of their own
        if re.search(r"match_geom_kernel<true|match_array_kernel<true|force_kernel<\w+, true", r["Kernel_Name"]):
            name += " (y_true)"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
names = [n for _, _, n in rows]
# steady part: from the middle third of the match kernels to the end
starts = [i for i, n in enumerate(names) if n == "match_geom_kernel"]
lo, hi = starts[len(starts) // 3], starts[-1]
part = rows[lo:hi]
span = part[-1][0] - part[0][0]
n_steps = len([1 for _, _, n in part if n == "match_geom_kernel"])
print("steps in window: %d   span per step: %.2f us" % (n_steps, span / n_steps / 1e3))
dur = {}
for s, e, n in part:
    dur.setdefault(n, []).append(e - s)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-22s n %4d  mean %8.2f us  per step %8.2f us" % (n, len(v), sum(v) / len(v) / 1e3, sum(v) / n_steps / 1e3))
ev = []
for s, e, n in part:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
level, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[min(level, 3)] = hist.get(min(level, 3), 0) + (t - last)
    last = t
    level += d
tot = sum(hist.values())
print("concurrency: " + "  ".join("%d%s kernels %.1f%%" % (k, "+" if k == 3 else "", 100.0 * v / tot) for k, v in sorted(hist.items())))
