#!/usr/bin/env python3
"""Isolated per-kernel times of the bench batch for one library build: profiles/kt.py [bench args]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse_args()
args.streams = 1
import torch
torch.cuda.set_device(0)
wl = bench.Workload(args, 0)
t = bench.stage_times(wl, 50)
print(os.environ.get("SSDBOX_LIB", "default").split("/")[-1], os.environ.get("SSDBOX_MATCH_GROUPS", "-"), {k: round(v, 2) for k, v in t.items()})
