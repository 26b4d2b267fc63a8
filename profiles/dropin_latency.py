#!/usr/bin/env python3
"""Wall time of the single-image drop-in functions (the reference's call pattern: one image per
call, numpy in, numpy out, so every call includes its host<->device copies) on the bench
geometry.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from face_detection_amd import synthetic as S                       # noqa: E402
from face_detection_amd.utils import anchor_utils, box_utils        # noqa: E402


def timeit(fn, reps=30):
    # first-use costs (allocations, code loading) stay outside, and so does the slow start of a
    # process that has just imported torch (the first few hundred milliseconds of calls run 5x slower)
    t_warm = time.perf_counter()
    n_warm = 0
    while n_warm < 40 or time.perf_counter() - t_warm < 0.5:
        fn()
        n_warm += 1
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    import torch
    cfg = S.reference_ssd_config(640, "config")
    rng = np.random.default_rng(5)
    boxes, labels, counts = S.make_ground_truth(rng, 1, 640, gmax=64)
    y = S.make_head_outputs(rng, cfg, boxes, counts)[0]
    g = int(counts[0])
    anchors = anchor_utils.generate_anchors(cfg)
    gt_b, gt_l = boxes[0, :g], labels[0, :g]
    out = {"anchors": int(anchors.shape[0]), "gt": g}
    out["generate_anchors_ms"] = timeit(lambda: anchor_utils.generate_anchors(cfg))
    out["compute_target_numpy_ms"] = timeit(lambda: box_utils.compute_target(anchors, gt_b, gt_l, 640.0))
    locs = np.ascontiguousarray(y[:, 2:])
    out["decode_numpy_ms"] = timeit(lambda: box_utils.decode(anchors, locs))
    dec = box_utils.decode(anchors, locs)
    m = y[:, 1] > 0.05
    cb, cs = dec[m], y[m, 1]
    out["candidates"] = int(m.sum())
    out["compute_nms_numpy_ms"] = timeit(lambda: box_utils.compute_nms(cb, cs, 0.45, 1000))
    # the same with device-resident torch tensors (no copies)
    d_locs = torch.from_numpy(locs).cuda()
    d_cb, d_cs = torch.from_numpy(cb).cuda(), torch.from_numpy(cs).cuda()
    d_gb, d_gl = torch.from_numpy(gt_b).cuda(), torch.from_numpy(gt_l).cuda()

    def sync(fn):
        def run():
            fn()
            torch.cuda.synchronize()
        return run
    out["compute_target_torch_ms"] = timeit(sync(lambda: box_utils.compute_target(anchors, d_gb, d_gl, 640.0)))
    out["decode_torch_ms"] = timeit(sync(lambda: box_utils.decode(anchors, d_locs)))
    out["compute_nms_torch_ms"] = timeit(sync(lambda: box_utils.compute_nms(d_cb, d_cs, 0.45, 1000)))
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}, default=str))


if __name__ == "__main__":
    main()
