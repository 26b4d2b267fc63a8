#!/usr/bin/env python3
"""The step with the order of its two independent halves alternating between the batch streams
(even streams: match+encode then decode+NMS; odd streams: the reverse) against the same order
on every stream.  profiles/order_mix.py [bench args]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args()
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
n = len(wl.slots)
def same(i):
    s = wl.slots[i % n]; wl.match(s); wl.detect(s)
def mixed(i):
    s = wl.slots[i % n]
    if (i % n) & 1:
        wl.detect(s); wl.match(s)
    else:
        wl.match(s); wl.detect(s)
def mixed2(i):
    s = wl.slots[i % n]
    if (i % n) >= n // 2:
        wl.detect(s); wl.match(s)
    else:
        wl.match(s); wl.detect(s)
for name, fn in (("same", same), ("alternating", mixed), ("halves", mixed2), ("same", same), ("alternating", mixed)):
    t = bench.timed_loop(torch, dist, 1, fn, args.steps, 20)
    print("%-12s %.3f M images/s" % (name, args.batch * args.steps / t / 1e6))
