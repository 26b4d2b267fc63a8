"""Three restatements of the reference's box math written independently of each other -- numpy
(oracle/box_oracle.py), scalar C (oracle/ssdbox_oracle.c) and PyTorch float32
(oracle/torch_oracle.py) -- agree: bit for bit wherever only IEEE + - * / min max and
comparisons are involved (anchors, IoU, matching, NMS), to a few ULP where exp / log come from
different math libraries (encode, decode).  None of them was compared with the reference itself
(parity unpinned); this test guards against an error of reading made in one of them."""
import numpy as np
import pytest
import torch

from conftest import ssd_config
from oracle import box_oracle as O
from oracle import c_oracle as C
from oracle import torch_oracle as T
from face_detection_amd import synthetic as S

F32 = np.float32
ODD = {'img_height': 96, 'img_width': 160, 'image_size': (96, 160), 'offsets': 0.37, 'n_classes': 1,
       'scales': [0.05, 0.2, 0.5, 0.9], 'aspect_ratios_per_layer': [[1.0, 2.0], [0.5, 1.0, 3.0], [1.0]],
       'two_boxes_for_ar1': True, 'steps': [8, 16, 32], 'variances': [0.1, 0.2]}


def bits(a):
    return np.ascontiguousarray(np.asarray(a, F32)).view(np.uint32)


def ulp_apart(a, b):
    """Largest distance in units of the last place between two float32 arrays (same sign pattern
    assumed where both are finite and non-zero; equal bit patterns count 0)."""
    a, b = np.asarray(a, F32), np.asarray(b, F32)
    both_nan = np.isnan(a) & np.isnan(b)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2 ** 31) - ib, ib)
    d = np.abs(ia - ib)
    d[both_nan] = 0
    return int(d.max()) if d.size else 0


@pytest.mark.parametrize("cfg", [ssd_config(64), ssd_config(128, "model"), ODD], ids=["64", "128-model", "odd"])
def test_anchors_agree(cfg):
    want = O.generate_anchors(cfg)
    got = T.generate_anchors(cfg).numpy()
    assert got.shape == want.shape
    assert np.array_equal(bits(got), bits(want))
    assert np.array_equal(bits(C.generate_anchors(cfg)), bits(want))


@pytest.mark.parametrize("seed", range(6))
def test_iou_match_and_targets_agree(seed):
    rng = np.random.default_rng(seed)
    cfg = [ssd_config(64), ssd_config(96, "model"), ODD][seed % 3]
    size = max(cfg['img_height'], cfg['img_width'])
    anchors = O.generate_anchors(cfg)
    boxes, labels, counts = S.make_ground_truth(rng, 2, size, gmax=40, stress=seed % 2 == 1)
    for b in range(2):
        g = int(counts[b])
        gt, lab = boxes[b, :g].copy(), labels[b, :g].copy()
        if seed == 4 and g > 3:
            gt[1] = gt[0]                                    # two identical boxes: forced-match order
            gt[2, 2:] = gt[2, :2]                            # a box of zero area
        corners = O.transform_center_to_corner(anchors)
        assert np.array_equal(bits(T.transform_center_to_corner(anchors)), bits(corners))
        assert np.array_equal(bits(T.transform_corner_to_center(gt)), bits(O.transform_corner_to_center(gt)))
        iou_np = O.compute_iou(corners, gt, float(size))
        iou_t = T.compute_iou(corners, gt, float(size)).numpy()
        assert np.array_equal(bits(iou_t), bits(iou_np))
        conf_t, locs_t, idx_t, best_t = T.compute_target(anchors, gt, lab, float(size), details=True)
        idx_np, best_np, _ = O.match(anchors, gt, float(size))
        assert np.array_equal(idx_t.numpy(), idx_np)
        assert np.array_equal(bits(best_t), bits(best_np))
        conf_np, locs_np = O.compute_target(anchors, gt, lab, float(size))
        assert np.array_equal(bits(conf_t), bits(conf_np))
        # x / y offsets are pure IEEE; the log-size offsets come from two math libraries
        assert np.array_equal(bits(locs_t[:, :2]), bits(locs_np[:, :2]))
        assert ulp_apart(locs_t[:, 2:].numpy(), locs_np[:, 2:]) <= 4
        conf_c, locs_c = C.compute_target(anchors, gt, lab, float(size))
        assert np.array_equal(bits(conf_c), bits(conf_np))
        assert np.array_equal(bits(locs_c[:, :2]), bits(locs_np[:, :2]))


@pytest.mark.parametrize("seed", range(4))
def test_decode_and_nms_agree(seed):
    rng = np.random.default_rng(100 + seed)
    cfg = ssd_config(96)
    anchors = O.generate_anchors(cfg)
    boxes, _, counts = S.make_ground_truth(rng, 1, 96, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=24)[0]
    dec_np = O.decode(anchors, y[:, 2:])
    dec_t = T.decode(anchors, y[:, 2:]).numpy()
    # exp from two libraries, then corner = centre -+ size / 2: an ULP of a size near 1 is 6e-8, and
    # the difference cancels, so the bound is absolute
    np.testing.assert_allclose(dec_t, dec_np, rtol=1e-6, atol=5e-7)
    # NMS on the same decoded boxes (ties, planted clusters): the same kept lists from all three
    scores = y[:, 1].copy()
    if seed == 1:
        scores[::7] = np.nan
        scores[3::11] = -0.0
        scores[5::11] = 0.0
    for thr, limit in ((0.45, 200), (0.0, 1000), (0.9, 50)):
        want = O.compute_nms(dec_np, scores, thr, limit)
        assert np.array_equal(T.compute_nms(dec_np, scores, thr, limit).numpy(), want)
        assert np.array_equal(C.compute_nms(dec_np, scores, thr, limit), want)
    # the whole inference tail, each with its own exp: kept anchors equal, rows to 1e-5
    d_np, i_np = O.postprocess(y, anchors, 0.05, 640.0, 480.0, 0.45, 1000)
    d_t, i_t = T.postprocess(y, anchors, 0.05, 640.0, 480.0, 0.45, 1000)
    assert np.array_equal(i_t.numpy(), i_np)
    np.testing.assert_allclose(d_t.numpy(), d_np, rtol=1e-5, atol=1e-4)


def test_empty_and_degenerate_nms():
    assert T.compute_nms(np.zeros((0, 4), F32), np.zeros((0,), F32), 0.45).shape == (0,)
    degenerate = np.array([[.1, .1, .1, .1], [.2, .2, .2, .2], [0, 0, .5, .5]], F32)
    s = np.array([.9, .8, .7], F32)
    assert list(T.compute_nms(degenerate, s, 0.45, 10).numpy()) == list(O.compute_nms(degenerate, s, 0.45, 10)) == [0, 2]
