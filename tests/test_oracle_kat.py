"""Hand-derived known-answer tests for the numpy oracle.

The reference has no tests or golden vectors for this path, so these values were derived
by hand from the reference source (utils/anchor_utils.py, utils/box_utils.py,
inference.py:33-83, utils/datasets.py:152-159); they pin the oracle's conventions, not the
reference's outputs (parity unpinned).
"""
import math

import numpy as np
import pytest

from oracle import box_oracle as O
from conftest import ssd_config

F32 = np.float32


def tiny_config(ratios, two_boxes=True, size=64, steps=(32,), scales=(0.5, 1.0)):
    return {'img_height': size, 'img_width': size, 'offsets': 0.5,
            'scales': list(scales), 'steps': list(steps),
            'aspect_ratios_per_layer': [list(r) for r in ratios],
            'two_boxes_for_ar1': two_boxes}


def test_k1_anchor_order_and_second_square():
    a = O.generate_anchors(tiny_config([[1.0]]))
    assert a.dtype == np.float32 and a.shape == (8, 4)
    s2 = F32(math.sqrt(0.5 * 1.0))
    expect = []
    for cy in (0.25, 0.75):          # rows (y) outer
        for cx in (0.25, 0.75):      # columns (x) inner
            expect.append([cx, cy, 0.5, 0.5])
            expect.append([cx, cy, s2, s2])
    np.testing.assert_array_equal(a, np.array(expect, F32))


def test_k2_non_unit_ratios_no_second_box_and_clip():
    cfg = tiny_config([[0.9, 1.8, 2.5]], scales=(0.8, 1.0))
    a = O.generate_anchors(cfg)
    assert a.shape == (12, 4)                       # 4 cells x 3, ratio == 1.0 never true
    r = math.sqrt(2.5)
    assert 0.8 * r > 1.0                            # clip is active on this width
    np.testing.assert_array_equal(a[2], np.array([0.25, 0.25, 1.0, F32(0.8 / r)], F32))
    np.testing.assert_array_equal(
        a[0], np.array([0.25, 0.25, F32(0.8 * math.sqrt(0.9)), F32(0.8 / math.sqrt(0.9))], F32))


def test_anchor_counts_match_table():
    # cells = sum((S // step)^2), A = cells * boxes per cell
    for size, a_cfg, a_model in ((512, 65280, 87040), (640, 102000, 136000)):
        cells = sum((size // s) ** 2 for s in (4, 8, 16, 32))
        assert cells * 3 == a_cfg and cells * 4 == a_model
    a = O.generate_anchors(ssd_config(64))
    assert a.shape == ((16 * 16 + 8 * 8 + 4 * 4 + 2 * 2) * 3, 4)
    a = O.generate_anchors(ssd_config(64, "model"))
    assert a.shape == ((16 * 16 + 8 * 8 + 4 * 4 + 2 * 2) * 4, 4)


def test_k3_iou_values():
    a = np.array([[0, 0, .5, .5]], F32)
    b = np.array([[.25, .25, .75, .75], [.6, .6, .9, .9], [0, 0, .5, .5]], F32)
    iou = O.compute_iou(a, b)
    assert iou.dtype == np.float32
    assert iou[0, 0] == F32(0.0625) / (F32(0.25) + F32(0.25) - F32(0.0625))   # 1/7
    assert iou[0, 1] == 0.0 and not np.signbit(iou[0, 1])
    assert iou[0, 2] == 1.0
    z = np.array([[.3, .3, .3, .3]], F32)                                     # zero area
    assert np.isnan(O.compute_iou(z, z)[0, 0])
    assert O.compute_iou(z, a)[0, 0] == 0.0


def test_area_clip_uses_max_size():
    tl = np.array([[0., 0.]], F32)
    br = np.array([[3., 700.]], F32)
    assert O.compute_area(tl, br)[0] == 3 * 512
    assert O.compute_area(tl, br, 1000.0)[0] == 2100
    assert O.compute_area(br, tl)[0] == 0


def grid_anchors():
    # 4 x 4 cells, one square box of side 0.25 per cell -> anchors tile the unit square
    cfg = tiny_config([[1.0]], two_boxes=False, size=64, steps=(16,), scales=(0.25, 0.5))
    return O.generate_anchors(cfg)


def test_k4_tiny_gt_forces_exactly_one_positive():
    anchors = grid_anchors()
    gt = np.array([[0.30, 0.30, 0.34, 0.34]], F32)       # inside cell (1,1): IoU = .0016/.0625
    labels = np.ones((1,), F32)
    conf, locs = O.compute_target(anchors, gt, labels)
    assert conf.sum() == 1 and conf[5] == 1              # cell (row 1, col 1) -> index 5
    expect = O.encode(anchors, np.repeat(gt, len(anchors), 0))
    np.testing.assert_array_equal(locs, expect)          # every negative encodes GT 0


def test_k5_duplicate_best_anchor_highest_gt_wins():
    anchors = grid_anchors()
    gt = np.array([[0.30, 0.30, 0.34, 0.34],
                   [0.28, 0.28, 0.36, 0.36]], F32)       # both best on anchor 5, both < 0.5
    labels = np.array([1., 1.], F32)
    idx, iou, bda = O.match(anchors, gt)
    assert list(bda) == [5, 5]
    assert idx[5] == 1 and iou[5] == 1.0
    conf, locs = O.compute_target(anchors, gt, labels)
    assert conf.sum() == 1
    np.testing.assert_array_equal(locs[5], O.encode(anchors[5:6], gt[1:2])[0])
    # all the other anchors have IoU 0 with both -> argmax tie -> GT 0
    assert (np.delete(idx, 5) == 0).all()


def test_k6_gt_overlapping_nothing_claims_anchor_zero():
    anchors = grid_anchors()
    gt = np.array([[0.55, 0.55, 0.70, 0.70],
                   [2.0, 2.0, 3.0, 3.0]], F32)           # second box is outside every anchor
    idx, iou, bda = O.match(anchors, gt)
    assert bda[1] == 0 and idx[0] == 1 and iou[0] == 1.0
    conf, _ = O.compute_target(anchors, gt, np.array([1., 1.], F32))
    assert conf[0] == 1


def test_threshold_is_not_less_than():
    anchors = np.array([[0.5, 0.5, 0.5, 0.5], [0.1, 0.1, 0.1, 0.1]], F32)   # corners .25-.75
    gt = np.array([[0.25, 0.25, 0.75, 0.5],                                 # IoU exactly 0.5
                   [0.05, 0.05, 0.15, 0.15]], F32)                          # claims anchor 1
    idx, iou, _ = O.match(anchors, gt)
    # anchor 0 is also GT 0's forced best; use a third anchor identical to 0 to see the rule
    anchors3 = np.concatenate([anchors, anchors[:1]], 0)
    idx, iou, bda = O.match(anchors3, gt)
    assert bda[0] == 0 and iou[2] == F32(0.5)
    conf, _ = O.compute_target(anchors3, gt, np.array([1., 1.], F32))
    assert conf[2] == 1                                   # 0.5 is not < 0.5 -> positive


def test_k7_encode_decode_round_trip():
    rng = np.random.default_rng(0)
    anchors = O.generate_anchors(ssd_config(64))
    lo = rng.uniform(0, 0.6, (len(anchors), 2)).astype(F32)
    wh = rng.uniform(0.02, 0.4, (len(anchors), 2)).astype(F32)
    boxes = np.concatenate([lo, lo + wh], 1)
    back = O.decode(anchors, O.encode(anchors, boxes))
    np.testing.assert_allclose(back, boxes, rtol=2e-5, atol=2e-6)


def test_encode_formula_divides_by_product():
    a = np.array([[0.5, 0.5, 0.3, 0.2]], F32)
    b = np.array([[0.4, 0.4, 0.7, 0.9]], F32)
    t = O.encode(a, b)[0]
    c = O.transform_corner_to_center(b)[0]
    assert t[0] == (c[0] - a[0, 0]) / (a[0, 2] * F32(0.1))
    assert t[3] == np.log(c[3] / a[0, 3]) / F32(0.2)


def nms_case():
    boxes = np.array([[0.0, 0.0, 0.5, 0.5],      # A
                      [0.0, 0.0, 0.5, 0.3],      # B: inside A, IoU = 0.6
                      [0.6, 0.6, 0.9, 0.9]], F32)  # C: disjoint
    scores = np.array([0.9, 0.8, 0.7], F32)
    return boxes, scores


@pytest.mark.parametrize("nms", [O.compute_nms, O.compute_nms_fast])
def test_k8_basic_suppression(nms):
    boxes, scores = nms_case()
    keep = nms(boxes, scores, 0.45, 1000)
    assert keep.dtype == np.int32 and list(keep) == [0, 2]
    perm = np.array([2, 0, 1])                    # indices refer to the input order
    assert list(nms(boxes[perm], scores[perm], 0.45, 1000)) == [1, 0]


@pytest.mark.parametrize("nms", [O.compute_nms, O.compute_nms_fast])
def test_k9_equal_scores_lower_index_first(nms):
    boxes, _ = nms_case()
    scores = np.array([0.5, 0.5, 0.5], F32)
    assert list(nms(boxes, scores, 0.45, 1000)) == [0, 2]
    assert list(nms(boxes[[1, 0, 2]], scores, 0.45, 1000)) == [0, 2]   # now B ranks first


@pytest.mark.parametrize("nms", [O.compute_nms, O.compute_nms_fast])
def test_k10_empty(nms):
    keep = nms(np.zeros((0, 4), F32), np.zeros((0,), F32), 0.45, 1000)
    assert keep.dtype == np.int32 and keep.shape == (0,)


@pytest.mark.parametrize("nms", [O.compute_nms, O.compute_nms_fast])
def test_k11_limit_truncates_before_suppression(nms):
    boxes, scores = nms_case()
    assert list(nms(boxes, scores, 0.45, 2)) == [0]        # C never enters
    assert list(nms(boxes, scores, 0.45, 1)) == [0]


@pytest.mark.parametrize("nms", [O.compute_nms, O.compute_nms_fast])
def test_k12_iou_equal_to_threshold_is_kept(nms):
    boxes = np.array([[0.0, 0.0, 0.5, 0.5], [0.0, 0.0, 0.5, 0.25]], F32)   # IoU exactly 0.5
    scores = np.array([0.9, 0.8], F32)
    assert list(nms(boxes, scores, 0.5, 10)) == [0, 1]
    assert list(nms(boxes, scores, np.nextafter(F32(0.5), F32(0)), 10)) == [0]


@pytest.mark.parametrize("nms", [O.compute_nms, O.compute_nms_fast])
def test_nms_degenerate_boxes_suppress_each_other(nms):
    # two zero-area boxes: IoU is 0/0 = NaN, NaN <= thr is false -> the later one dies;
    # a zero-area box against a real one has IoU 0 -> both stay
    boxes = np.array([[.1, .1, .1, .1], [.2, .2, .2, .2], [0, 0, .5, .5]], F32)
    scores = np.array([0.9, 0.8, 0.7], F32)
    assert list(nms(boxes, scores, 0.45, 10)) == [0, 2]


def test_nms_fast_equals_literal_on_random_clusters():
    rng = np.random.default_rng(3)
    for n in (1, 2, 37, 300):
        c = rng.uniform(0.2, 0.8, (max(n // 6, 1), 2))
        ctr = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.01, (n, 2))
        wh = rng.uniform(0.05, 0.1, (n, 2))
        boxes = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(F32)
        scores = rng.uniform(0, 1, n).astype(F32)
        scores[: n // 3] = F32(1.0)
        for limit in (5, 200):
            np.testing.assert_array_equal(O.compute_nms(boxes, scores, 0.45, limit),
                                          O.compute_nms_fast(boxes, scores, 0.45, limit))


def test_postprocess_layout_strict_threshold_and_scaling():
    anchors = grid_anchors()
    y = np.zeros((16, 6), F32)
    y[:, 0] = 0.95
    y[:, 1] = 0.05                         # equal to the threshold -> rejected (strict >)
    y[5, :2] = (0.1, 0.9)
    y[10, :2] = (0.4, 0.6)
    dets, idx = O.postprocess(y, anchors, 0.05, img_width=200, img_height=100)
    assert list(idx) == [5, 10] and dets.shape == (2, 6)
    box5 = O.transform_center_to_corner(anchors[5:6])[0]
    np.testing.assert_array_equal(
        dets[0], np.array([1, F32(0.9), box5[0] * 200, box5[1] * 100, box5[2] * 200, box5[3] * 100], F32))
    dets0, idx0 = O.postprocess(y, anchors, 0.0)        # evaluation.py:67 -> everything passes
    assert len(idx0) == 16                              # the grid boxes only touch -> IoU 0


def test_y_true_layout():
    anchors = grid_anchors()
    gt = np.array([[0.30, 0.30, 0.34, 0.34]], F32)
    y = O.build_y_true(anchors, gt, np.ones((1,), F32), 64.0)
    assert y.shape == (16, 6) and y.dtype == np.float32
    assert list(y[5, :2]) == [0, 1] and list(y[4, :2]) == [1, 0]
    np.testing.assert_array_equal(y[:, 2:], O.encode(anchors, np.repeat(gt, 16, 0)))


def test_softmax2_hand_values_and_c_restatement():
    """models/ssd.py:95 for two classes: equal logits give exactly 0.5 each, a gap of 200
    saturates to exactly 0 / 1, the location columns pass through; C equals numpy to 2 ULP."""
    from oracle import c_oracle as C
    rows = np.array([[0, 0, 1, 2, 3, 4], [-100, 100, 0, 0, 0, 0], [5, 5, 0, 0, 0, 0],
                     [0, np.log(3.0), 0, 0, 0, 0], [-3.5, 1.25, 9, 9, 9, 9]], np.float32)
    for sm in (O.softmax2(rows), C.softmax2(rows)):
        assert sm.dtype == np.float32 and sm.shape == rows.shape
        assert sm[0, :2].tolist() == [0.5, 0.5] and sm[2, :2].tolist() == [0.5, 0.5]
        assert sm[1, :2].tolist() == [0.0, 1.0]
        np.testing.assert_allclose(sm[3, :2], [0.25, 0.75], rtol=3e-7)
        np.testing.assert_array_equal(sm[:, 2:], rows[:, 2:])
    rng = np.random.default_rng(3)
    y = rng.normal(0, 4, (2, 500, 6)).astype(np.float32)
    np.testing.assert_allclose(C.softmax2(y), O.softmax2(y), rtol=3e-7, atol=1e-37)
