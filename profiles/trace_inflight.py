#!/usr/bin/env python3
"""Timeline of several batches in flight without a profiler attached (under rocprofv3's kernel
trace, and with an event pair on every launch, the launches of different streams no longer
overlap as they do in the timed loops).  Needs the experiment build of the library, whose kernels
stamp their own begin and end on the device's wall clock:

  make -C face_detection_amd/csrc BUILD=build_trace OUT=$PWD/face_detection_amd/libssdbox_trace.so EXTRA=-DSSD_EXP_TRACE
  SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_trace.so python3 profiles/trace_inflight.py \\
      [--half decode|match|step] [bench args: --streams N --steps K ...]

Prints, for the steady middle of the run: wall time per batch, every kernel's duration under
load, the gaps inside a chain, the chain length (first kernel's begin to last kernel's end of one
call), and how much of the time 0 / 1 / 2 / ... streaming kernels (select, match) were running."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

half = "decode"
if "--half" in sys.argv:
    i = sys.argv.index("--half")
    half = sys.argv[i + 1]
    del sys.argv[i:i + 2]
args = bench.parse_args()
import torch
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
fn = {"step": wl.step, "match": lambda i: wl.match(*wl.pick(i)), "decode": lambda i: wl.detect(*wl.pick(i))}[half]
n = len(wl.slots)
for i in range(200):
    fn(i)
torch.cuda.synchronize()
import ctypes
lib = wl.lib
lib.ssdbox_debug_trace_reset.restype = ctypes.c_int
lib.ssdbox_debug_trace_dump.restype = ctypes.c_int
lib.ssdbox_debug_trace_dump.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int]
lib.ssdbox_debug_trace_reset()
steps = min(args.steps, 2000)
import time
t_0 = time.perf_counter()
for i in range(steps):
    fn(i)
t_enq = time.perf_counter() - t_0
torch.cuda.synchronize()
t_all = time.perf_counter() - t_0
print("host: enqueue %.2f us per call, wall %.2f us per call" % (t_enq / steps * 1e6, t_all / steps * 1e6))
cap = 1 << 18
import numpy as np
st = np.zeros(cap, np.int32); cid = np.zeros(cap, np.uint64); tb = np.zeros(cap, np.uint64); te = np.zeros(cap, np.uint64)
nrec = lib.ssdbox_debug_trace_dump(st.ctypes.data, cid.ctypes.data, tb.ctypes.data, te.ctypes.data, cap)
from face_detection_amd import _lib as L
ids = {int(s["ctx"].handle.value if hasattr(s["ctx"].handle, "value") else s["ctx"].handle): q for q, s in enumerate(wl.slots)}
t_org = int(tb[:nrec][tb[:nrec] > 0].min())
rows = []
for i in range(nrec):
    if tb[i] == 0 or te[i] == 0:
        continue
    rows.append(((int(tb[i]) - t_org) * 0.01, (int(te[i]) - t_org) * 0.01, L.STAGE_NAMES[st[i]], ids.get(int(cid[i]), -1)))
rows.sort()
first = "match" if half in ("match", "step") else "select"
last = "force" if half == "match" else "nms_scan"
idx = [i for i, r in enumerate(rows) if r[2] == first]
lo, hi = idx[len(idx) // 4], idx[3 * len(idx) // 4]
part = rows[lo:hi]
nb = len([1 for r in part if r[2] == first])
span = part[-1][0] - part[0][0]
print("half %s  streams %d  batches in window %d  wall per batch %.2f us  -> %.3fe6 images/s"
      % (half, n, nb, span / nb, args.batch / (span / nb)))
dur = collections.defaultdict(list)
for t0, t1, name, q in part:
    dur[name].append(t1 - t0)
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("  %-10s n %4d  mean %6.2f  median %6.2f  p10 %6.2f  p90 %6.2f us"
          % (name, len(v), sum(v) / len(v), v[len(v) // 2], v[len(v) // 10], v[int(len(v) * 0.9)]))
byq = collections.defaultdict(list)
for r in part:
    byq[r[3]].append(r)
gaps = collections.defaultdict(list)
chains = []
for q, v in byq.items():
    start = None
    for a, b in zip(v, v[1:]):
        gaps[a[2] + " -> " + b[2]].append(b[0] - a[1])
    for r in v:
        if r[2] == first:
            start = r[0]
        if r[2] == last and start is not None:
            chains.append(r[1] - start)
            start = None
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 4:
        v = sorted(v)
        print("  gap %-22s n %4d  mean %6.2f  median %6.2f  p90 %6.2f us" % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * 0.9)]))
if chains:
    chains.sort()
    print("  chain (begin of %s to end of %s): mean %.2f  median %.2f us" % (first, last, sum(chains) / len(chains), chains[len(chains) // 2]))
ev = []
for t0, t1, name, q in part:
    if name in ("select", "match"):
        ev.append((t0, 1))
        ev.append((t1, -1))
ev.sort()
level, prev, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[min(level, 4)] += t - prev
    prev = t
    level += d
tot = sum(hist.values())
print("  streaming kernels running: " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
if os.environ.get("INFLIGHT_DUMP"):
    t_base = part[0][0]
    for t0, t1, name, q in part[:80]:
        print("    q%d %-9s %8.2f -> %8.2f  (%5.2f)" % (q, name, t0 - t_base, t1 - t_base, t1 - t0))
