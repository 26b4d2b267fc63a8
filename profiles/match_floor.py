#!/usr/bin/env python3
"""What of match+encode's time is the box walk: the match half with four batches in flight and
the kernel alone, on the bench batch, on the same batch cut to its first k boxes per image, and
with one tiny box per image (no tile lists anything but box 0's tile):
   profiles/match_floor.py [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
args = bench.parse_args()
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
n = len(wl.slots)
full = wl.gt_counts.copy()


def run(name):
    wl.warm_all()
    r = []
    for _ in range(3):
        t = bench.timed_loop(torch, dist, 1, lambda i: wl.match(wl.slots[i % n]), args.steps, 10)
        r.append(t / args.steps * 1e6)
    st = bench.stage_times(wl, 30)
    print("%-28s boxes %5d  half %s us per batch   kernel alone: match %.2f force %.2f us"
          % (name, int(wl.d_counts.sum().item()), [round(x, 2) for x in r], st["match"], st.get("force", 0.0)))


run("bench batch")
for k in (8, 2, 1):
    wl.d_counts = torch.from_numpy(np.minimum(full, k).astype(np.int32)).cuda()
    run("first %d boxes" % k)
tiny = wl.gt_boxes.copy()
tiny[:, 0] = np.array([0.001, 0.001, 0.003, 0.003], np.float32)
wl.d_boxes = torch.from_numpy(tiny).cuda()
wl.d_counts = torch.ones_like(wl.d_counts)
run("one tiny box")
