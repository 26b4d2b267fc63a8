#!/usr/bin/env python3
"""How much of the bench step is host launch time: enqueue time vs total time for the same
steps, then the same work replayed from one captured graph per slot."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--no-cpu-baseline"] + sys.argv[1:]
import torch

import bench

args = bench.parse_args()
wl = bench.Workload(args, 0)
steps = 400
for i in range(40):
    wl.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    wl.step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("eager: enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))

graphs = []
for slot in wl.slots:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=slot["stream"]):
        wl.match(slot)
        wl.detect(slot)
    graphs.append(g)
torch.cuda.synchronize()
def replay(i):
    # (replay() launches on torch's current stream: each graph goes to its slot's stream)
    with torch.cuda.stream(wl.slots[i % len(graphs)]["stream"]):
        graphs[i % len(graphs)].replay()


for i in range(40):
    replay(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    replay(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("graph: enqueue %.1f us/step, total %.1f us/step" % ((t1 - t0) / steps * 1e6, (t2 - t0) / steps * 1e6))
