#!/usr/bin/env python3
"""Kernels of several batches in flight, from a rocprofv3 --kernel-trace CSV: per kernel the
duration under load (not alone on the chip), per queue the gap between consecutive kernels,
the wall time per batch and how much of the time the streaming kernels (the ones that use HBM)
were running: chain_trace.py DIR [select|match]

The steady part of the trace is the middle half of the launches of the named kernel."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "select"
rows = []
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        m = re.search(r"(\w+)_kernel", r["Kernel_Name"])
        name = m.group(1) if m else r["Kernel_Name"][:24]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if r[2].startswith(anchor)]
if len(idx) < 8:
    sys.exit("fewer than 8 launches of %s*" % anchor)
lo, hi = idx[len(idx) // 4], idx[3 * len(idx) // 4]
part = rows[lo:hi]
n_batches = len([1 for r in part if r[2].startswith(anchor)])
span = part[-1][0] - part[0][0]
print("batches in window: %d   wall per batch: %.2f us" % (n_batches, span / n_batches / 1e3))
dur = collections.defaultdict(list)
for s, e, n, q in part:
    dur[n].append(e - s)
for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("  %-16s n %4d  mean %7.2f  median %7.2f  p90 %7.2f us" %
          (n, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[int(len(v) * 0.9)] / 1e3))
# gaps inside a queue: end of a kernel to the start of the next one on the same queue
byq = collections.defaultdict(list)
for r in part:
    byq[r[3]].append(r)
gaps = collections.defaultdict(list)
for q, v in byq.items():
    for a, b in zip(v, v[1:]):
        gaps[a[2] + " -> " + b[2]].append(b[0] - a[1])
print("queues: %d" % len(byq))
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 4:
        v = sorted(v)
        print("  gap %-28s n %4d  mean %6.2f  median %6.2f us" % (k, len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3))
# how many streaming kernels were running at any time
ev = []
for s, e, n, q in part:
    if n.startswith(("select", "match")):
        ev.append((s, 1))
        ev.append((e, -1))
ev.sort()
level, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(level, 4)] += t - last
    last = t
    level += d
tot = sum(hist.values())
print("streaming kernels running: " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
