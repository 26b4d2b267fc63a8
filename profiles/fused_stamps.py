#!/usr/bin/env python3
"""Phase times of the fused top-k + NMS kernel for image 0 of the bench batch (library built with
make EXTRA=-DSSD_EXP_STAMPS: the kernel prints its own clock stamps):
  SSDBOX_LIB=.../libssdbox_stamps.so python3 profiles/fused_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args()
import torch
torch.cuda.set_device(0)
args.streams = 1
wl = bench.Workload(args, 0)
for _ in range(4):
    wl.detect(wl.slots[0])
    torch.cuda.synchronize()
