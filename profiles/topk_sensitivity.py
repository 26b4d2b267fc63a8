#!/usr/bin/env python3
"""Top-k kernel time against the make-up of the candidates (tie group, planted clusters,
threshold): which part of the kernel the bench batch exercises."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch                                                        # noqa: E402
from face_detection_amd import _lib, synthetic as S                 # noqa: E402
from face_detection_amd.runtime import Context, geometry_for        # noqa: E402


def main():
    lib = _lib.load()
    cfg = S.reference_ssd_config(640, "config")
    dev = torch.device("cuda", 0)
    geom = geometry_for(cfg, 0)
    A, B = geom.num_anchors, 32
    stream = torch.cuda.Stream(device=dev)
    ctx = Context(0, stream)
    dets = torch.empty((B, 1000, 6), dtype=torch.float32, device=dev)
    keep = torch.empty((B, 1000), dtype=torch.int32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for n_ties, dup, thr in ((64, 5, 0.05), (0, 5, 0.05), (0, 0, 0.05), (64, 5, 0.03), (64, 5, 0.08), (1000, 5, 0.05)):
        rng = np.random.default_rng(1000)
        boxes, _, counts = S.make_ground_truth(rng, B, 640, gmax=256)
        y = torch.from_numpy(S.make_head_outputs(rng, cfg, boxes, counts, n_ties=n_ties, dup_per_gt=dup)).to(dev)
        n_cand = int((y[..., 1] > thr).sum().item()) // B

        def run():
            _lib.check(lib.ssdbox_decode_nms(ctx.handle, geom.handle, None, A, p(y), B, thr, 0.45, 1000, None,
                                             p(dets), p(keep), p(count)))
        for _ in range(5):
            run()
        ctx.synchronize()
        ctx.set_timing(True)
        for _ in range(30):
            run()
        t = ctx.collect_timing()
        ctx.set_timing(False)
        print("ties %4d dup %d thr %.2f cand/img %5d:" % (n_ties, dup, thr, n_cand),
              {k: round(ms / n * 1e3, 2) for k, (ms, n) in t.items() if n})


if __name__ == "__main__":
    main()
