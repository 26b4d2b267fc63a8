#!/usr/bin/env python3
"""The step with the two halves of a batch on different streams: M streams take the match+encode
calls, D streams the decode+NMS calls (the halves are independent of each other).
profiles/split_streams.py M D [bench args]"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M, D = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0]] + sys.argv[3:] + ["--streams", str(M + D)]
import bench
args = bench.parse_args()
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
ms, ds = wl.slots[:M], wl.slots[M:]
def step(i):
    wl.match(ms[i % M])
    wl.detect(ds[i % D])
res = []
for rep in range(3):
    t = bench.timed_loop(torch, dist, 1, step, args.steps, 10)
    res.append(round(args.batch * args.steps / t / 1e6, 3))
print("match streams %d, detect streams %d: step" % (M, D), res)
