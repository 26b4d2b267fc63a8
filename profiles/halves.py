#!/usr/bin/env python3
"""Throughput of the two halves and of the step with N batches in flight, several repetitions:
profiles/halves.py [bench args]   (no CPU baseline, no parity)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args()
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
n = len(wl.slots)
res = {"match": [], "decode": [], "step": []}
for rep in range(3):
    t = bench.timed_loop(torch, dist, 1, lambda i: wl.match(wl.slots[i % n]), args.steps, 10)
    res["match"].append(args.batch * args.steps / t)
    t = bench.timed_loop(torch, dist, 1, lambda i: wl.detect(wl.slots[i % n]), args.steps, 10)
    res["decode"].append(args.batch * args.steps / t)
    t = bench.timed_loop(torch, dist, 1, wl.step, args.steps, 10)
    res["step"].append(args.batch * args.steps / t)
print(os.environ.get("SSDBOX_DBG_SKIP", "-"), {k: [round(x / 1e6, 3) for x in v] for k, v in res.items()})
