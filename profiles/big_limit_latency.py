#!/usr/bin/env python3
"""Latency of decode+NMS on the general path (more than 1024 ranked boxes per image) next to the
fast path, batch 32 @640x640, device-resident inputs:  profiles/big_limit_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = bench.parse_args()
import torch
torch.cuda.set_device(0)
import face_detection_amd as fda
from face_detection_amd.utils import anchor_utils
wl = bench.Workload(args, 0)
grid = anchor_utils.generate_anchors(wl.cfg)
y = wl.sets[0]["ypred"] if hasattr(wl, "sets") else wl.d_ypred
for thr in (0.05, 0.0):
    for limit in (1000, 1024, 1025, 2000, 4096, 4097, 8192, 20000):
        for _ in range(3):
            out = fda.decode_nms_batch(grid, y, thr, 0.45, limit)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = fda.decode_nms_batch(grid, y, thr, 0.45, limit)
        torch.cuda.synchronize()
        print("thr %.2f limit %5d: %.3f ms per batch of %d, kept %d" % (thr, limit, (time.perf_counter() - t0) / 10 * 1e3,
                                                                      args.batch, int(out[2].sum())))
