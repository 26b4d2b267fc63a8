#!/usr/bin/env python3
"""Phase times of topk_sort_kernel for image 0 of the bench batch (library built with
make EXTRA=-DSSD_EXP_STAMPS: the kernel prints its own clock stamps):  profiles/topk_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args()
import torch
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
for _ in range(6):
    wl.detect(wl.slots[0])
    torch.cuda.synchronize()
