#!/usr/bin/env python3
"""Regenerates tests/golden/box_pipeline_small.npz.

The reference ships no golden vectors for this path and its runtime (TensorFlow 2.4) is not
installed here, so these vectors come from this repository's numpy oracle
(oracle/box_oracle.py, the literal restatement of reference utils/anchor_utils.py and
utils/box_utils.py): they freeze the oracle's conventions against regressions, they do not
pin the reference's outputs (parity unpinned).  Inputs are seeded; run from the repo root:

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import box_oracle as O                     # noqa: E402
from face_detection_amd import synthetic as S          # noqa: E402

CONFIG = {
    'img_height': 96, 'img_width': 128, 'offsets': 0.5, 'two_boxes_for_ar1': True,
    'scales': [0.05, 0.15, 0.4, 0.9], 'steps': [8, 16, 32],
    'aspect_ratios_per_layer': [[0.9, 1.8, 2.5], [1.0, 2.0, 0.5], [1.0, 3.0]],
}


def main():
    rng = np.random.default_rng(2024)
    anchors = O.generate_anchors(CONFIG)
    out = {"anchors": anchors}
    boxes, labels, counts = S.make_ground_truth(rng, 3, 96, gmax=12)
    labels[1, :counts[1]] = np.arange(1, counts[1] + 1)            # labels other than 1
    out.update(gt_boxes=boxes, gt_labels=labels, gt_counts=counts)
    confs, locs, idxs, ious = [], [], [], []
    for b in range(3):
        g = counts[b]
        idx, iou, _ = O.match(anchors, boxes[b, :g], 128.0)
        c, l = O.compute_target(anchors, boxes[b, :g], labels[b, :g], 128.0)
        confs.append(c), locs.append(l), idxs.append(idx.astype(np.int32)), ious.append(iou)
    out.update(conf=np.stack(confs), locs=np.stack(locs), match_idx=np.stack(idxs), match_iou=np.stack(ious))
    # head outputs: scores with ties, clusters around the boxes
    A = len(anchors)
    y = np.empty((3, A, 6), np.float32)
    y[..., 2:] = rng.standard_normal((3, A, 4)).astype(np.float32) * 0.5
    face = rng.uniform(0, 1, (3, A)).astype(np.float32) ** 4
    face[:, ::37] = 1.0
    face[:, 5::53] = 0.25
    y[..., 1] = face
    y[..., 0] = 1 - face
    out["y_pred"] = y
    for thr_name, thr in (("t05", 0.05), ("t00", 0.0)):
        dets = np.zeros((3, 1000, 6), np.float32)
        keep = np.full((3, 1000), -1, np.int32)
        cnt = np.zeros((3,), np.int32)
        for b in range(3):
            d, k = O.postprocess(y[b], anchors, thr, 640.0, 480.0, 0.45, 1000)
            dets[b, :len(k)], keep[b, :len(k)], cnt[b] = d, k, len(k)
        out["dets_" + thr_name], out["keep_" + thr_name], out["count_" + thr_name] = dets[:, :A], keep[:, :A], cnt
        assert cnt.max() <= A
    # raw NMS
    n = 300
    ctr = rng.uniform(0.2, 0.8, (20, 2))[rng.integers(0, 20, n)] + rng.normal(0, 0.01, (n, 2))
    wh = rng.uniform(0.04, 0.12, (n, 2))
    nb = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(np.float32)
    ns = rng.uniform(0, 1, n).astype(np.float32)
    ns[::9] = 0.5
    out.update(nms_boxes=nb, nms_scores=ns, nms_keep_200=O.compute_nms(nb, ns, 0.45, 200),
               nms_keep_20=O.compute_nms(nb, ns, 0.45, 20))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "box_pipeline_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; anchors", A)


if __name__ == "__main__":
    main()
