#!/usr/bin/env python3
"""Halves and step with the batch streams at different priorities (does an uneven share of the
chip keep the chains out of step?):  profiles/prio_mix.py [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args()
import torch
import torch.distributed as dist
from face_detection_amd.runtime import Context
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
n = len(wl.slots)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
for name, prios in (("all high", [-1] * n), ("all normal", [0] * n), ("alternating", [-1, 0] * (n // 2)),
                    ("one high", [-1] + [0] * (n - 1)), ("one normal", [0] + [-1] * (n - 1))):
    for slot, p in zip(wl.slots, prios):
        slot["stream"] = torch.cuda.Stream(device=0, priority=p)
        slot["ctx"] = Context(0, slot["stream"])
    wl.warm_all()
    res = {}
    for key, fn in (("match", lambda i: wl.match(wl.slots[i % n])), ("decode", lambda i: wl.detect(wl.slots[i % n])),
                    ("step", wl.step)):
        r = [args.batch * args.steps / bench.timed_loop(torch, dist, 1, fn, args.steps, 10) / 1e6 for _ in range(3)]
        res[key] = [round(x, 3) for x in r]
    print("%-12s %s" % (name, res))
