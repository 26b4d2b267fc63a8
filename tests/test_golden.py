"""Committed golden vectors (tests/golden/box_pipeline_small.npz, made by
tests/golden/make_golden.py from the numpy oracle -- the reference has none for this path):
the C oracle on CPU and the HIP path on the GPU must reproduce them -- integers exactly,
floats to 1e-5 relative (the golden floats went through numpy's exp/log)."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import box_oracle as O
from oracle import c_oracle as C

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "box_pipeline_small.npz"))

_spec = importlib.util.spec_from_file_location(
    "make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)
CONFIG = _mod.CONFIG

RTOL, ATOL = 1e-5, 1e-6


def test_golden_file_is_what_the_generator_makes():
    np.testing.assert_array_equal(O.generate_anchors(CONFIG), GOLD["anchors"])
    b = 1
    g = GOLD["gt_counts"][b]
    conf, locs = O.compute_target(GOLD["anchors"], GOLD["gt_boxes"][b, :g], GOLD["gt_labels"][b, :g], 128.0)
    np.testing.assert_array_equal(conf, GOLD["conf"][b])
    np.testing.assert_array_equal(locs, GOLD["locs"][b])


def test_c_oracle_matches_golden():
    anchors = C.generate_anchors(CONFIG)
    np.testing.assert_array_equal(anchors, GOLD["anchors"])
    for b in range(3):
        g = GOLD["gt_counts"][b]
        conf, locs, idx, iou, _ = C.compute_target(anchors, GOLD["gt_boxes"][b, :g], GOLD["gt_labels"][b, :g],
                                                   128.0, details=True)
        np.testing.assert_array_equal(idx, GOLD["match_idx"][b])
        np.testing.assert_array_equal(iou, GOLD["match_iou"][b])
        np.testing.assert_array_equal(conf, GOLD["conf"][b])
        np.testing.assert_allclose(locs, GOLD["locs"][b], rtol=RTOL, atol=ATOL)
        for name, thr in (("t05", 0.05), ("t00", 0.0)):
            dets, keep = C.postprocess(GOLD["y_pred"][b], anchors, thr, 640.0, 480.0, 0.45, 1000)
            n = GOLD["count_" + name][b]
            assert len(keep) == n
            np.testing.assert_array_equal(keep, GOLD["keep_" + name][b, :n])
            np.testing.assert_allclose(dets, GOLD["dets_" + name][b, :n], rtol=RTOL, atol=1e-3)
    for lim in (200, 20):
        np.testing.assert_array_equal(C.compute_nms(GOLD["nms_boxes"], GOLD["nms_scores"], 0.45, lim),
                                      GOLD["nms_keep_%d" % lim])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["geom", "array"])
def test_hip_path_matches_golden(mode):
    import face_detection_amd as fda
    from face_detection_amd.utils import anchor_utils, box_utils
    grid = anchor_utils.generate_anchors(CONFIG)
    np.testing.assert_array_equal(np.asarray(grid), GOLD["anchors"])
    anchors = grid if mode == "geom" else GOLD["anchors"]
    conf, locs, idx, iou = fda.match_encode_batch(anchors, GOLD["gt_boxes"], GOLD["gt_counts"], GOLD["gt_labels"],
                                                  128.0, return_match=True)
    np.testing.assert_array_equal(idx, GOLD["match_idx"])
    np.testing.assert_array_equal(iou, GOLD["match_iou"])
    np.testing.assert_array_equal(conf, GOLD["conf"])
    np.testing.assert_allclose(locs, GOLD["locs"], rtol=RTOL, atol=ATOL)
    wh = np.tile(np.array([[640.0, 480.0]], np.float32), (3, 1))
    for name, thr in (("t05", 0.05), ("t00", 0.0)):
        dets, keep, count = fda.decode_nms_batch(anchors, GOLD["y_pred"], thr, 0.45, 1000, wh)
        np.testing.assert_array_equal(count, GOLD["count_" + name])
        A = GOLD["keep_" + name].shape[1]
        np.testing.assert_array_equal(keep[:, :A], GOLD["keep_" + name])
        np.testing.assert_allclose(dets[:, :A], GOLD["dets_" + name], rtol=RTOL, atol=1e-3)
    for lim in (200, 20):
        np.testing.assert_array_equal(box_utils.compute_nms(GOLD["nms_boxes"], GOLD["nms_scores"], 0.45, lim),
                                      GOLD["nms_keep_%d" % lim])
