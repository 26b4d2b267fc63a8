"""The data formats either side of the box pipeline: ground-truth preparation (reference
utils/datasets.py:131-140, utils/transforms.py:28-33 and :60-63) and WiderFace result rows
(reference evaluation.py:67-79).  The reference has no fixtures for them: the oracle is pinned
by hand-derived values, the HIP path is compared with the oracle bit for bit."""
import os

import numpy as np
import pytest

from oracle import box_oracle as O

F32 = np.float32


def test_prepare_gt_hand_values():
    # 1000 x 500 image, box (101, 51, 200, 100): corners (101,51,301,151) - 1 -> /(1000,500)
    # = (.1,.1,.3,.3); h/w = .5 < 1 -> y' = (y - .5) / (640*1000) * (500*640) + .5
    b = O.prepare_gt([[101, 51, 200, 100]], 1000, 500, 640, 640)
    np.testing.assert_allclose(b, [[0.1, 0.3, 0.3, 0.4]], rtol=0, atol=1e-7)
    # tall image: x is squeezed instead
    b = O.prepare_gt([[51, 101, 100, 200]], 500, 1000, 640, 640)
    np.testing.assert_allclose(b, [[0.3, 0.1, 0.4, 0.3]], rtol=0, atol=1e-7)
    # square image into a square input: the last step is the identity up to rounding
    b = O.prepare_gt([[11, 21, 30, 40]], 100, 100, 320, 320)
    np.testing.assert_allclose(b, [[0.1, 0.2, 0.4, 0.6]], rtol=0, atol=1e-7)
    # shrink to half size placed at offset (h_off, w_off) = (100, 200) in a 1000 x 1000 image:
    # x' = (x * 500 + 200 - 1) / 1000
    b = O.prepare_gt([[101, 101, 200, 200]], 1000, 1000, 640, 640, shrink=(500, 500, 100, 200))
    np.testing.assert_allclose(b, [[0.249, 0.149, 0.349, 0.249]], rtol=0, atol=1e-6)
    assert b.dtype == np.float32


def test_widerface_rows_hand_values():
    preds = np.array([[1, 0.98765, 10.9, 20.2, 30.99, 50.5], [1, 0.5, -0.5, 0.0, 3.2, 4.9]], F32)
    boxes, scores = O.widerface_rows(preds)
    # truncation toward zero (also for the negative corner), +1, then w = x1 - x0
    assert boxes.tolist() == [[11, 21, 20, 30], [1, 1, 3, 4]]
    assert O.widerface_lines("img_7", preds) == "img_7\n2\n11 21 20 30 0.988\n1 1 3 4 0.500\n"
    assert O.widerface_lines("none", np.zeros((0, 6), F32)) == "none\n0\n"


def test_format_matches_oracle_text(tmp_path):
    from face_detection_amd.utils import eval_utils
    preds = np.array([[1, 0.98765, 10.9, 20.2, 30.99, 50.5], [1, 0.5, 7.0, 8.0, 9.0, 11.0]], F32)
    boxes, scores = O.widerface_rows(preds)
    assert eval_utils.format_widerface_result("a", boxes, scores) == O.widerface_lines("a", preds)
    pad = np.concatenate([boxes, np.zeros((3, 4), boxes.dtype)])
    pads = np.concatenate([scores, np.zeros(3)])
    paths = eval_utils.write_widerface_results(str(tmp_path / "res"), ["a", "b"], [pad, pad], [pads, pads], [2, 0])
    assert open(paths[0]).read() == O.widerface_lines("a", preds)
    assert open(paths[1]).read() == "b\n0\n"
    assert os.path.basename(paths[1]) == "b.txt"


def _random_annotations(rng, B, G):
    wh = np.stack([rng.integers(60, 3000, B), rng.integers(60, 3000, B)], 1).astype(np.int32)
    counts = rng.integers(1, G + 1, B).astype(np.int32)
    raw = np.zeros((B, G, 4), F32)
    for b in range(B):
        n = counts[b]
        raw[b, :n, 0] = rng.integers(0, wh[b, 0] - 20, n)
        raw[b, :n, 1] = rng.integers(0, wh[b, 1] - 20, n)
        raw[b, :n, 2] = rng.integers(1, 400, n)
        raw[b, :n, 3] = rng.integers(1, 400, n)
    return raw, counts, wh


@pytest.mark.gpu
@pytest.mark.parametrize("target", [(640, 640), (300, 512), (1024, 768)])
def test_prepare_gt_batch_bit_exact(target):
    import face_detection_amd as fda
    rng = np.random.default_rng(11)
    B, G = 37, 70
    raw, counts, wh = _random_annotations(rng, B, G)
    wh[0] = (target[1], target[0])           # aspect ratio equal to the target's
    wh[1] = (2 * target[1], 2 * target[0])
    shrink = np.zeros((B, 4), np.int32)
    for b in range(0, B, 2):                 # every other image was shrunk
        h2 = rng.integers(wh[b, 1] // 2, wh[b, 1] + 1)
        w2 = rng.integers(wh[b, 0] // 2, wh[b, 0] + 1)
        shrink[b] = (h2, w2, rng.integers(0, wh[b, 1] - h2 + 1), rng.integers(0, wh[b, 0] - w2 + 1))
    for sh in (None, shrink):
        got = fda.prepare_gt_batch(raw, counts, wh, target, sh)
        assert got.dtype == np.float32 and got.shape == (B, G, 4)
        for b in range(B):
            n = counts[b]
            s = tuple(sh[b]) if sh is not None and sh[b, 0] > 0 else None
            want = O.prepare_gt(raw[b, :n], wh[b, 0], wh[b, 1], target[0], target[1], s)
            np.testing.assert_array_equal(got[b, :n].view(np.uint32), want.view(np.uint32))
            assert not got[b, n:].any()
    with pytest.raises(ValueError):
        fda.prepare_gt_batch(raw, counts, np.zeros((B, 2), np.int32), target)


@pytest.mark.gpu
def test_prepared_boxes_feed_match_encode():
    """annotation -> prepare -> match on the GPU equals the oracle chain image by image."""
    import face_detection_amd as fda
    from conftest import ssd_config
    from oracle import c_oracle as C
    cfg = ssd_config(128)
    rng = np.random.default_rng(5)
    B, G = 6, 12
    raw, counts, wh = _random_annotations(rng, B, G)
    boxes = fda.prepare_gt_batch(raw, counts, wh, (128, 128))
    from face_detection_amd.utils import anchor_utils
    grid = anchor_utils.generate_anchors(cfg)
    conf, locs = fda.match_encode_batch(grid, boxes, counts, None, 128.0)
    anchors = O.generate_anchors(cfg)
    for b in range(B):
        n = counts[b]
        want = O.prepare_gt(raw[b, :n], wh[b, 0], wh[b, 1], 128, 128)
        c_conf, c_locs = C.compute_target(anchors, want, np.ones((n,), F32), 128.0)
        np.testing.assert_array_equal(conf[b], c_conf)
        np.testing.assert_array_equal(locs[b].view(np.uint32), c_locs.view(np.uint32))


@pytest.mark.gpu
def test_widerface_rows_batch_exact():
    import face_detection_amd as fda
    from face_detection_amd.utils import eval_utils
    rng = np.random.default_rng(3)
    B, limit = 9, 300
    count = rng.integers(0, limit + 1, B).astype(np.int32)
    count[0], count[1] = 0, limit
    dets = np.zeros((B, limit, 6), F32)
    for b in range(B):
        n = count[b]
        x0 = rng.uniform(-3, 1900, n)
        y0 = rng.uniform(-3, 1000, n)
        dets[b, :n] = np.stack([np.ones(n), rng.uniform(0, 1, n), x0, y0, x0 + rng.uniform(0, 300, n),
                                y0 + rng.uniform(0, 300, n)], 1)
    xywh, scores = fda.widerface_rows_batch(dets, count)
    assert xywh.dtype == np.int32 and xywh.shape == (B, limit, 4)
    for b in range(B):
        n = count[b]
        want, w_scores = O.widerface_rows(dets[b, :n])
        np.testing.assert_array_equal(xywh[b, :n], want)
        np.testing.assert_array_equal(scores[b, :n], w_scores.astype(F32))
        assert not xywh[b, n:].any() and not scores[b, n:].any()
        assert eval_utils.format_widerface_result("f%d" % b, xywh[b], scores[b], n) == \
            O.widerface_lines("f%d" % b, dets[b, :n])
