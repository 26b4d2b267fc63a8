"""GPU parity: the HIP path, called through the C ABI, against the oracles on the same
seeded inputs.  Integer results (match indices, masks, kept indices, counts) must be equal;
float results must be bit-identical to the scalar C oracle (same IEEE operation sequence,
exp/log included) and within 1e-5 relative of the numpy oracle (libm-quality exp/log).
"""
import os
import threading

import numpy as np
import pytest

from conftest import ssd_config
from oracle import box_oracle as O
from oracle import c_oracle as C
from face_detection_amd import synthetic as S

pytestmark = pytest.mark.gpu
# other synthetic batches for the full-size cases (a longer hunt): SSDBOX_TEST_SEED_OFFSET=1, 2, ...
SEED_OFFSET = int(os.environ.get("SSDBOX_TEST_SEED_OFFSET", "0"))
F32 = np.float32


@pytest.fixture(scope="module")
def fda():
    import face_detection_amd as pkg
    from face_detection_amd.utils import anchor_utils, box_utils
    pkg.anchor_utils = anchor_utils
    pkg.box_utils = box_utils
    return pkg


def bits(a):
    return np.ascontiguousarray(a, F32).view(np.uint32)


def assert_bits_equal(got, want):
    got, want = np.asarray(got, F32), np.asarray(want, F32)
    assert got.shape == want.shape
    same = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want)) | ((got == 0) & (want == 0))
    if not same.all():
        bad = np.argwhere(~same)[:5]
        raise AssertionError("float bits differ at %s: got %s want %s"
                             % (bad.tolist(), got[~same][:5], want[~same][:5]))


ODD_CONFIG = {'img_height': 100, 'img_width': 75, 'offsets': 0.3, 'two_boxes_for_ar1': False,
              'scales': [0.1, 0.35, 0.9, 1.4], 'steps': [8, 16, 30],
              'aspect_ratios_per_layer': [[1.0, 3.0], [0.5], [1.0, 2.0, 0.5, 3.0]]}


def test_selftest_building_blocks(fda):
    rng = np.random.default_rng(0)
    x_log = np.concatenate([np.exp(rng.uniform(-87, 88, 3000)), rng.uniform(0.5, 2, 3000),
                            [0.0, -1.0, np.inf, 1e-42, 1.0, np.nan]]).astype(F32)
    x_exp = np.concatenate([rng.uniform(-104, 89, 3000), rng.uniform(-1, 1, 3002),
                            [0.0, 100.0, -200.0, np.nan]]).astype(F32)
    fda.default_context().selftest(x_log, C.logf(x_log), x_exp, C.expf(x_exp))


@pytest.mark.parametrize("cfg", [ssd_config(64), ssd_config(64, "model"), ssd_config(640),
                                 ssd_config(512, "model"), ODD_CONFIG])
def test_generate_anchors_bit_exact(fda, cfg):
    grid = fda.anchor_utils.generate_anchors(cfg)
    want = O.generate_anchors(cfg)
    assert grid.shape == want.shape and len(grid) == len(want)
    assert_bits_equal(np.asarray(grid), want)
    assert_bits_equal(grid[5:9, 2:], want[5:9, 2:])


def check_match(fda, cfg, boxes, labels, counts, max_size, mode):
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg) if mode == "geom" else anchors_np
    conf, locs, midx, miou = fda.match_encode_batch(grid, boxes, counts, labels, max_size,
                                                    return_match=True)
    ytrue = fda.match_encode_batch(grid, boxes, counts, labels, max_size, y_true=True)
    for b in range(len(counts)):
        g = counts[b]
        c_conf, c_locs, c_idx, c_iou, _ = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g],
                                                           max_size, details=True)
        np.testing.assert_array_equal(midx[b], c_idx)
        assert_bits_equal(miou[b], c_iou)
        np.testing.assert_array_equal(conf[b], c_conf)
        assert_bits_equal(locs[b], c_locs)
        assert_bits_equal(ytrue[b], C.y_true(c_conf, c_locs))


@pytest.mark.parametrize("mode", ["geom", "array"])
@pytest.mark.parametrize("seed,size,ratios,stress,gmax", [
    (0, 128, "config", False, 32), (1, 128, "model", True, 40), (2, 256, "config", True, 256),
    (3, 96, "config", True, 300), (4, 64, "model", False, 1),
    # the 512 x 512 image of the plumbing configuration (A = 65 280 / 87 040)
    (5, 512, "config", False, 256), (6, 512, "model", False, 256)])
def test_match_encode_small(fda, mode, seed, size, ratios, stress, gmax):
    rng = np.random.default_rng(seed)
    cfg = ssd_config(size, ratios)
    boxes, labels, counts = S.make_ground_truth(rng, 4, size, gmax=gmax, stress=stress)
    check_match(fda, cfg, boxes, labels, counts, float(size), mode)


def test_match_encode_odd_geometry_and_numpy_oracle(fda):
    rng = np.random.default_rng(9)
    boxes, labels, counts = S.make_ground_truth(rng, 3, 75, gmax=20, stress=True)
    check_match(fda, ODD_CONFIG, boxes, labels, counts, 100.0, "geom")
    check_match(fda, ODD_CONFIG, boxes, labels, counts, 100.0, "array")
    anchors_np = O.generate_anchors(ODD_CONFIG)
    conf, locs = fda.match_encode_batch(anchors_np, boxes, counts, labels, 100.0)
    for b in range(3):
        n_conf, n_locs = O.compute_target(anchors_np, boxes[b], labels[b], 100.0)
        np.testing.assert_array_equal(conf[b], n_conf)
        np.testing.assert_allclose(locs[b], n_locs, rtol=1e-5, atol=1e-6)


def test_match_edge_cases(fda):
    cfg = ssd_config(64)
    grid = fda.anchor_utils.generate_anchors(cfg)
    anchors_np = O.generate_anchors(cfg)
    gt = np.zeros((4, 4, 4), F32)
    lab = np.ones((4, 4), F32)
    gt[0, :2] = [[0.30, 0.30, 0.34, 0.34], [0.28, 0.28, 0.36, 0.36]]   # shared best anchor
    gt[1, :2] = [[0.55, 0.55, 0.70, 0.70], [2.0, 2.0, 3.0, 3.0]]       # box outside everything
    gt[2, :3] = [[0.1, 0.1, 0.4, 0.5]] * 3                              # identical boxes
    gt[3, :1] = [[0.0, 0.0, 1.0, 1.0]]                                  # whole image
    counts = np.array([2, 2, 3, 1], np.int32)
    lab[2, :3] = [1, 2, 3]
    for anchors in (grid, anchors_np):
        conf, locs, midx, miou = fda.match_encode_batch(anchors, gt, counts, lab, 64.0,
                                                        return_match=True)
        for b in range(4):
            g = counts[b]
            c_conf, c_locs, c_idx, c_iou, _ = C.compute_target(anchors_np, gt[b, :g], lab[b, :g],
                                                               64.0, details=True)
            np.testing.assert_array_equal(midx[b], c_idx)
            np.testing.assert_array_equal(conf[b], c_conf)
            assert_bits_equal(miou[b], c_iou)
            assert_bits_equal(locs[b], c_locs)
    assert midx[1][0] == 1 and conf[1][0] == 1          # untouched box claims anchor 0
    # thresholds at and below zero make every anchor a positive, also where no box is near
    for thr in (0.0, -1.0, 0.9999):
        for anchors in (grid, anchors_np):
            conf, locs = fda.match_encode_batch(anchors, gt, counts, lab, 64.0, iou_threshold=thr)
            for b in range(4):
                g = counts[b]
                c_conf, c_locs = C.compute_target(anchors_np, gt[b, :g], lab[b, :g], 64.0, thr)
                np.testing.assert_array_equal(conf[b], c_conf)
                assert_bits_equal(locs[b], c_locs)
    with pytest.raises(ValueError):
        fda.match_encode_batch(grid, gt, np.array([2, 0, 3, 1], np.int32), lab, 64.0)
    with pytest.raises(ValueError):
        fda.box_utils.compute_target(grid, np.zeros((0, 4), F32), np.zeros((0,), F32))


def test_match_encode_large_batch_image_order(fda):
    """Images are processed in order of falling box count up to 4096 images per call and as
    given beyond that: both sides of the limit, very uneven counts, results per image unchanged."""
    cfg = ssd_config(64)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    for B in (4096, 4100):
        rng = np.random.default_rng(B)
        boxes, labels, counts = S.make_ground_truth(rng, B, 64, gmax=40)
        conf, locs = fda.match_encode_batch(grid, boxes, counts, labels, 64.0)
        for b in list(rng.choice(B, 12, replace=False)) + [0, B - 1, int(np.argmax(counts)), int(np.argmin(counts))]:
            g = counts[b]
            c_conf, c_locs = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g], 64.0)
            np.testing.assert_array_equal(conf[b], c_conf)
            assert_bits_equal(locs[b], c_locs)


def test_match_encode_full_size_batch32(fda):
    rng = np.random.default_rng(21 + SEED_OFFSET)
    cfg = ssd_config(640)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    boxes, labels, counts = S.make_ground_truth(rng, 32, 640, gmax=256)
    boxes[0], labels[0], counts[0] = S.make_ground_truth(rng, 1, 640, gmax=256, stress=True)[0][0], 1.0, 256
    conf, locs, midx, miou = fda.match_encode_batch(grid, boxes, counts, labels, 640.0,
                                                    return_match=True)
    assert conf.shape == (32, 102000) and locs.shape == (32, 102000, 4)
    for b in range(32):
        g = counts[b]
        c_conf, c_locs, c_idx, c_iou, ba = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g],
                                                            640.0, details=True)
        np.testing.assert_array_equal(midx[b], c_idx)
        np.testing.assert_array_equal(conf[b], c_conf)
        assert_bits_equal(locs[b], c_locs)
        # size-independent properties: every box owns its best anchor unless a later box took it
        owner = {}
        for gi, a in enumerate(ba):
            owner[a] = gi
        for a, gi in owner.items():
            assert midx[b][a] == gi and miou[b][a] == 1.0 and conf[b][a] == 1.0
        assert ((conf[b] == 1) == ~(miou[b] < 0.5)).all()


def test_match_1024_dense_anchor_case(fda):
    # the reference's anchor out-of-memory case: 1024^2, batch 16, nothing of size (A,G)
    rng = np.random.default_rng(33 + SEED_OFFSET)
    cfg = ssd_config(1024)
    grid = fda.anchor_utils.generate_anchors(cfg)
    anchors_np = O.generate_anchors(cfg)
    boxes, labels, counts = S.make_ground_truth(rng, 16, 1024, gmax=256)
    ytrue = fda.match_encode_batch(grid, boxes, counts, labels, 1024.0, y_true=True)
    assert ytrue.shape == (16, 261120, 6)
    assert (ytrue[:, :, :2].sum(-1) == 1).all()
    for b in (0, 7, 15):
        g = counts[b]
        c_conf, c_locs = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g], 1024.0)
        assert_bits_equal(ytrue[b], C.y_true(c_conf, c_locs))


def check_decode_nms(fda, cfg, y, thr, img_wh, anchors_mode="geom", limit=1000):
    anchors_np = O.generate_anchors(cfg)
    anchors = fda.anchor_utils.generate_anchors(cfg) if anchors_mode == "geom" else anchors_np
    dets, keep, count = fda.decode_nms_batch(anchors, y, thr, 0.45, limit, img_wh)
    assert dets.shape == (len(y), limit, 6) and keep.shape == (len(y), limit)
    for b in range(len(y)):
        w, h = (img_wh[b] if img_wh is not None else (1.0, 1.0))
        c_dets, c_idx = C.postprocess(y[b], anchors_np, thr, float(w), float(h), 0.45, limit)
        n = int(count[b])
        assert n == len(c_idx)
        np.testing.assert_array_equal(keep[b, :n], c_idx)
        assert_bits_equal(dets[b, :n], c_dets)
        assert (keep[b, n:] == -1).all() and (dets[b, n:] == 0).all()
    return dets, keep, count


@pytest.mark.parametrize("mode", ["geom", "array"])
@pytest.mark.parametrize("thr", [0.05, 0.3, 0.0])
def test_decode_nms_small(fda, mode, thr):
    rng = np.random.default_rng(5)
    cfg = ssd_config(128)
    boxes, _, counts = S.make_ground_truth(rng, 5, 128, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=40)
    wh = rng.uniform(100, 2000, (5, 2)).astype(F32)
    check_decode_nms(fda, cfg, y, thr, wh, mode)
    check_decode_nms(fda, cfg, y, thr, None, mode, limit=200)
    check_decode_nms(fda, cfg, y, thr, None, mode, limit=7)


@pytest.mark.parametrize("ratios", ["config", "model"])
def test_decode_nms_one_512_image(fda, ratios):
    """One 512 x 512 image (the plumbing configuration) through decode + NMS, both anchor sets."""
    rng = np.random.default_rng(21)
    cfg = ssd_config(512, ratios)
    boxes, _, counts = S.make_ground_truth(rng, 1, 512, gmax=256)
    y = S.make_head_outputs(rng, cfg, boxes, counts)
    check_decode_nms(fda, cfg, y, 0.05, np.array([[1920.0, 1080.0]], F32), "geom")
    check_decode_nms(fda, cfg, y, 0.0, None, "array")


@pytest.mark.parametrize("thr", [0.0, 0.01, -1.0])
def test_decode_nms_many_candidates(fda, thr):
    """Thresholds below the usual ones (evaluation.py:67 uses 0): nearly every anchor is a
    candidate and the candidates are pruned by a grid-wide histogram before the top-k."""
    rng = np.random.default_rng(21)
    cfg = ssd_config(256)
    boxes, _, counts = S.make_ground_truth(rng, 3, 256, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=300)
    y[1, :, 1] = 0.25                       # one image where every score ties
    y[2, 100:, 1] = 0.0                     # one with fewer candidates than the limit at thr 0
    assert y.shape[1] > 8192
    check_decode_nms(fda, cfg, y, thr, None)
    check_decode_nms(fda, cfg, y, thr, None, "array", limit=200)
    check_decode_nms(fda, cfg, y, thr, None, limit=7)


def test_decode_nms_against_numpy_oracle(fda):
    rng = np.random.default_rng(6)
    cfg = ssd_config(128)
    anchors_np = O.generate_anchors(cfg)
    boxes, _, counts = S.make_ground_truth(rng, 3, 128, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=40)
    dets, keep, count = fda.decode_nms_batch(fda.anchor_utils.generate_anchors(cfg), y, 0.05)
    for b in range(3):
        n_dets, n_idx = O.postprocess(y[b], anchors_np, 0.05)          # the literal O(K^3) loop
        n = int(count[b])
        np.testing.assert_array_equal(keep[b, :n], n_idx)
        np.testing.assert_allclose(dets[b, :n], n_dets, rtol=1e-5, atol=1e-6)


def test_decode_nms_from_logits(fda):
    """Class columns before the softmax of models/ssd.py:95: same detections, bit for bit, as
    the softmaxed rows (C oracle's softmax) through the ordinary entry point; even anchor
    counts take the 16-byte select kernel, odd ones the strided one."""
    rng = np.random.default_rng(16)
    cfg = ssd_config(128)
    anchors_np = O.generate_anchors(cfg)
    boxes, _, counts = S.make_ground_truth(rng, 4, 128, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=40)
    p1 = np.clip(y[..., 1].astype(np.float64), 1e-7, 1 - 1e-7)
    logits = y.copy()
    logits[..., 0] = rng.normal(0, 2, p1.shape)
    logits[..., 1] = logits[..., 0] + np.log(p1 / (1 - p1))
    logits[0, :7, :2] = [[0, 0], [-100, 100], [100, -100], [3, 3], [-0.0, 0.0], [88, 88], [-200, -200]]
    soft = C.softmax2(logits)
    np.testing.assert_allclose(soft, O.softmax2(logits), rtol=2e-6, atol=1e-30)
    assert soft[0, 0, 1] == 0.5 and soft[0, 1, 1] == 1.0 and soft[0, 2, 1] == 0.0
    wh = rng.uniform(100, 2000, (4, 2)).astype(F32)
    for anchors, yl, ys in ((fda.anchor_utils.generate_anchors(cfg), logits, soft),
                            (anchors_np, logits, soft),
                            (anchors_np[:-1], logits[:, :-1], soft[:, :-1])):
        for thr in (0.05, 0.0):
            got = fda.decode_nms_batch(anchors, yl, thr, 0.45, 1000, wh, from_logits=True)
            want = fda.decode_nms_batch(anchors, ys, thr, 0.45, 1000, wh)
            for g, w in zip(got, want):
                assert_bits_equal(g, w) if g.dtype == np.float32 else np.testing.assert_array_equal(g, w)
            assert int(got[2].sum()) > 0
    # and against the C oracle's post-processing of the softmaxed rows
    c_dets, c_idx = C.postprocess(soft[1], anchors_np, 0.05, float(wh[1, 0]), float(wh[1, 1]), 0.45, 1000)
    dets, keep, count = fda.decode_nms_batch(anchors_np, logits, 0.05, 0.45, 1000, wh, from_logits=True)
    np.testing.assert_array_equal(keep[1, :int(count[1])], c_idx)
    assert_bits_equal(dets[1, :int(count[1])], c_dets)


def test_decode_nms_nothing_passes_and_everything_ties(fda):
    cfg = ssd_config(64)
    A = len(O.generate_anchors(cfg))
    y = np.zeros((3, A, 6), F32)
    y[0, :, :2] = (0.99, 0.01)                  # nothing above 0.05
    y[1, :, :2] = (0.0, 1.0)                    # every score 1.0f: order is by anchor index
    y[2, :, :2] = (0.95, 0.05)                  # exactly the threshold: strict > rejects
    y[2, 17, 1] = np.nextafter(F32(0.05), F32(1))
    dets, keep, count = check_decode_nms(fda, cfg, y, 0.05, None)
    assert list(count[[0, 2]]) == [0, 1] and keep[2, 0] == 17


@pytest.mark.parametrize("thr", [0.05, 0.0])
def test_decode_nms_full_size_batch32(fda, thr):
    rng = np.random.default_rng(40 + SEED_OFFSET)
    cfg = ssd_config(640)
    boxes, _, counts = S.make_ground_truth(rng, 32, 640, gmax=256)
    y = S.make_head_outputs(rng, cfg, boxes, counts)
    dets, keep, count = check_decode_nms(fda, cfg, y, thr, None)
    anchors_np = O.generate_anchors(cfg)
    for b in range(32):
        n = int(count[b])
        assert 0 < n <= 1000
        s = dets[b, :n, 1]
        assert (np.diff(s) <= 0).all()                         # kept order is score order
        kb = dets[b, :n, 2:]
        iou = O.compute_iou(kb, kb)                            # survivors never overlap > thr
        np.fill_diagonal(iou, 0)
        raw = C.decode(anchors_np[keep[b, :n]], y[b, keep[b, :n], 2:])
        iou_raw = O.compute_iou(raw, raw)
        np.fill_diagonal(iou_raw, 0)
        assert (iou_raw <= F32(0.45)).all()
        assert (y[b, keep[b, :n], 1] > thr).all()


def clustered_boxes(rng, n):
    c = rng.uniform(0.2, 0.8, (max(n // 8, 1), 2))
    ctr = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.01, (n, 2))
    wh = rng.uniform(0.04, 0.1, (n, 2))
    return np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(F32)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 700, 1024, 1500, 2048, 2049, 5000, 8192, 8193, 12000, 70000])
def test_compute_nms_raw(fda, n):
    rng = np.random.default_rng(n)
    boxes = clustered_boxes(rng, n)
    scores = rng.uniform(0, 1, n).astype(F32)
    scores[rng.integers(0, n, n // 4)] = F32(1.0)
    for limit in (10, 200, 1000):
        got = fda.box_utils.compute_nms(boxes, scores, 0.45, limit)
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, C.compute_nms(boxes, scores, 0.45, limit))


def test_compute_nms_conventions(fda):
    nms = fda.box_utils.compute_nms
    boxes = np.array([[0, 0, .5, .5], [0, 0, .5, .3], [.6, .6, .9, .9]], F32)
    assert list(nms(boxes, np.array([.9, .8, .7], F32), 0.45, 1000)) == [0, 2]
    assert list(nms(boxes[[2, 0, 1]], np.array([.7, .9, .8], F32), 0.45, 1000)) == [1, 0]
    assert list(nms(boxes, np.array([.5, .5, .5], F32), 0.45, 1000)) == [0, 2]       # ties: index
    assert list(nms(boxes, np.array([.9, .8, .7], F32), 0.45, 2)) == [0]             # limit first
    empty = nms(np.zeros((0, 4), F32), np.zeros((0,), F32), 0.45, 1000)
    assert empty.dtype == np.int32 and empty.shape == (0,)
    half = np.array([[0, 0, .5, .5], [0, 0, .5, .25]], F32)                          # IoU == 0.5
    assert list(nms(half, np.array([.9, .8], F32), 0.5, 10)) == [0, 1]
    assert list(nms(half, np.array([.9, .8], F32), float(np.nextafter(F32(.5), F32(0))), 10)) == [0]
    degenerate = np.array([[.1, .1, .1, .1], [.2, .2, .2, .2], [0, 0, .5, .5]], F32)
    assert list(nms(degenerate, np.array([.9, .8, .7], F32), 0.45, 10)) == [0, 2]
    # all scores equal and more candidates than the sort buffer: ranks are the indices
    rng = np.random.default_rng(1)
    many = clustered_boxes(rng, 6000)
    same = np.full((6000,), 0.25, F32)
    np.testing.assert_array_equal(nms(many, same, 0.45, 300), C.compute_nms(many, same, 0.45, 300))
    # negative, zero and signed-zero scores, NaN last
    s = np.array([-1.0, 0.0, -0.0, np.nan, 3.0, -2.5], F32)
    b = np.array([[i * .1, 0, i * .1 + .05, .05] for i in range(6)], F32)
    assert list(nms(b, s, 0.45, 10)) == [4, 1, 2, 0, 5, 3]
    with pytest.raises(ValueError):
        nms(boxes, np.array([.9, .8, .7], F32), 1.0, 10)
    # all scores equal and more ranks than one launch of the general path takes (4096)
    np.testing.assert_array_equal(nms(many, same, 0.45, 5000), C.compute_nms(many, same, 0.45, 5000))


@pytest.mark.parametrize("n,limit", [(1025, 4096), (3000, 2000), (6000, 3000), (6000, 4096), (20000, 4096)])
def test_compute_nms_more_than_1024_ranks(fda, n, limit):
    """compute_nms takes any limit (utils/box_utils.py:184,202); above 1024 ranked boxes the
    library runs its general path (sort in LDS, the reference's greedy loop): same kept lists as
    the oracle, with clusters, exact score ties and degenerate boxes."""
    nms = fda.box_utils.compute_nms
    rng = np.random.default_rng(n + limit)
    boxes = clustered_boxes(rng, n)
    scores = rng.uniform(0, 1, n).astype(F32)
    scores[rng.integers(0, n, n // 8)] = F32(0.5)                 # ties: the lower index ranks first
    boxes[rng.integers(0, n, 20), 2] = boxes[rng.integers(0, n, 20), 0]
    for thr in (0.45, 0.0, -1.0):
        np.testing.assert_array_equal(nms(boxes, scores, thr, limit), C.compute_nms(boxes, scores, thr, limit))
    same = np.full((n,), 0.25, F32)
    np.testing.assert_array_equal(nms(boxes, same, 0.45, limit), C.compute_nms(boxes, same, 0.45, limit))


@pytest.mark.parametrize("n,limit", [(4097, 4097), (6000, 5000), (9000, 8192), (9000, 8193), (20000, 20000),
                                     (13000, 50000)])
def test_compute_nms_more_than_4096_ranks(fda, n, limit):
    """Beyond 4096 ranked boxes the general path takes the ranks 4096 at a time, the kept boxes of
    the earlier ranks suppressing the later ones first: the reference's loop, any limit."""
    nms = fda.box_utils.compute_nms
    rng = np.random.default_rng(n + limit)
    boxes = clustered_boxes(rng, n)
    boxes[:, :2] += rng.uniform(-0.15, 0.15, (n, 2)).astype(F32)   # spread: thousands of boxes survive
    boxes[:, 2:] = boxes[:, :2] + rng.uniform(0.01, 0.06, (n, 2)).astype(F32)
    scores = rng.uniform(0, 1, n).astype(F32)
    scores[rng.integers(0, n, n // 8)] = F32(0.5)                 # ties across the 4096-rank boundaries
    boxes[rng.integers(0, n, 20), 2] = boxes[rng.integers(0, n, 20), 0]
    for thr in (0.45, 0.0):
        want = C.compute_nms(boxes, scores, thr, limit)
        np.testing.assert_array_equal(nms(boxes, scores, thr, limit), want)
    assert len(C.compute_nms(boxes, scores, 0.45, limit)) > 1024
    same = np.full((n,), 0.25, F32)
    np.testing.assert_array_equal(nms(boxes, same, 0.45, limit), C.compute_nms(boxes, same, 0.45, limit))
    # and a smaller call on the same context afterwards
    np.testing.assert_array_equal(nms(boxes[:3000], scores[:3000], 0.45, 1000),
                                  C.compute_nms(boxes[:3000], scores[:3000], 0.45, 1000))


def test_decode_nms_more_than_4096_ranks(fda):
    rng = np.random.default_rng(37)
    cfg = ssd_config(256)
    boxes, _, counts = S.make_ground_truth(rng, 3, 256, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=300)
    wh = rng.uniform(100, 2000, (3, 2)).astype(F32)
    check_decode_nms(fda, cfg, y, 0.0, wh, limit=10000)            # every anchor a candidate, three launches
    check_decode_nms(fda, cfg, y, 0.0, None, "array", limit=30000)  # limit above the anchor count
    check_decode_nms(fda, cfg, y, 0.05, None, limit=5000)          # fewer candidates than 4096: later launches idle
    check_decode_nms(fda, cfg, y, 0.05, wh, limit=1000)


def test_decode_nms_more_than_1024_ranks(fda):
    rng = np.random.default_rng(31)
    cfg = ssd_config(256)
    boxes, _, counts = S.make_ground_truth(rng, 3, 256, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=300)
    wh = rng.uniform(100, 2000, (3, 2)).astype(F32)
    check_decode_nms(fda, cfg, y, 0.0, wh, limit=2500)
    check_decode_nms(fda, cfg, y, 0.01, None, "array", limit=4096)
    check_decode_nms(fda, cfg, y, 0.05, None, limit=1025)
    # and the fast path again on the same context (the ranked-box buffers are laid out wider now)
    check_decode_nms(fda, cfg, y, 0.05, wh, limit=1000)


def test_compute_nms_mask_paths(fda):
    """The suppression mask has a listed-pairs path and a pair-by-pair path (dense tiles,
    boxes of zero area, non-finite corners, negative threshold): both against the oracle."""
    nms = fda.box_utils.compute_nms
    rng = np.random.default_rng(77)
    # one tight cluster: every pair of a 64 x 64 tile overlaps
    ctr = np.array([[0.5, 0.5]]) + rng.normal(0, 0.01, (700, 2))
    wh = rng.uniform(0.2, 0.3, (700, 2))
    dense = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(F32)
    scores = rng.uniform(0, 1, 700).astype(F32)
    for thr in (0.45, 0.9, 0.0):
        np.testing.assert_array_equal(nms(dense, scores, thr, 1000), C.compute_nms(dense, scores, thr, 1000))
    # mixed: sparse boxes with some of zero area, inverted corners, NaN and infinite corners
    boxes = clustered_boxes(rng, 900)
    boxes[5] = [0.3, 0.3, 0.3, 0.6]
    boxes[70] = [0.4, 0.4, 0.2, 0.2]
    boxes[200, 2] = np.nan
    boxes[333] = [0.1, 0.1, np.inf, 0.5]
    boxes[640] = [-np.inf, 0.2, 0.4, 0.5]
    scores = rng.uniform(0, 1, 900).astype(F32)
    np.testing.assert_array_equal(nms(boxes, scores, 0.45, 1000), C.compute_nms(boxes, scores, 0.45, 1000))
    # negative threshold: a pair that does not overlap at all suppresses too
    sparse = clustered_boxes(rng, 300)
    s3 = rng.uniform(0, 1, 300).astype(F32)
    np.testing.assert_array_equal(nms(sparse, s3, -0.5, 1000), C.compute_nms(sparse, s3, -0.5, 1000))


def test_elementwise_functions(fda):
    bu = fda.box_utils
    rng = np.random.default_rng(2)
    cfg = ssd_config(64)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    A = len(anchors_np)
    lo = rng.uniform(0, 0.7, (A, 2))
    boxes = np.concatenate([lo, lo + rng.uniform(0.01, 0.3, (A, 2))], 1).astype(F32)
    assert_bits_equal(bu.transform_corner_to_center(boxes), O.transform_corner_to_center(boxes))
    assert_bits_equal(bu.transform_center_to_corner(anchors_np), O.transform_center_to_corner(anchors_np))
    assert_bits_equal(bu.transform_center_to_corner(grid), O.transform_center_to_corner(anchors_np))
    assert_bits_equal(bu.compute_area(boxes[:, :2], boxes[:, 2:]), O.compute_area(boxes[:, :2], boxes[:, 2:]))
    assert_bits_equal(bu.compute_area(boxes[:, :2] * 900, boxes[:, 2:] * 2000),
                      O.compute_area(boxes[:, :2] * 900, boxes[:, 2:] * 2000))
    b2 = boxes[:301]
    assert_bits_equal(bu.compute_iou(boxes, b2, 64.0), O.compute_iou(boxes, b2, 64.0))
    for anc in (grid, anchors_np):
        assert_bits_equal(bu.encode(anc, boxes), C.encode(anchors_np, boxes))
        np.testing.assert_allclose(bu.encode(anc, boxes), O.encode(anchors_np, boxes), rtol=1e-5, atol=1e-5)
        locs = rng.standard_normal((3, A, 4)).astype(F32)
        assert_bits_equal(bu.decode(anc, locs), C.decode(anchors_np, locs))
        np.testing.assert_allclose(bu.decode(anc, locs), O.decode(anchors_np, locs), rtol=1e-5, atol=1e-6)
        assert_bits_equal(bu.decode(anc, locs[0]), C.decode(anchors_np, locs[0]))
    conf, locs = bu.compute_target(grid, boxes[:7], np.ones((7,), F32), 64.0)
    c_conf, c_locs = C.compute_target(anchors_np, boxes[:7], np.ones((7,), F32), 64.0)
    np.testing.assert_array_equal(conf, c_conf)
    assert_bits_equal(locs, c_locs)


def test_torch_tensors_in_torch_tensors_out(fda):
    import torch
    cfg = ssd_config(64)
    grid = fda.anchor_utils.generate_anchors(cfg)
    anchors_np = O.generate_anchors(cfg)
    gt = torch.tensor([[0.2, 0.2, 0.5, 0.6]], device="cuda")
    conf, locs = fda.box_utils.compute_target(grid, gt, torch.ones(1, device="cuda"), 64.0)
    assert isinstance(conf, torch.Tensor) and conf.is_cuda and locs.shape == (len(anchors_np), 4)
    c_conf, c_locs = C.compute_target(anchors_np, gt.cpu().numpy(), np.ones(1, F32), 64.0)
    np.testing.assert_array_equal(conf.cpu().numpy(), c_conf)
    assert_bits_equal(locs.cpu().numpy(), c_locs)


def test_six_threads_each_with_a_context(fda):
    # the reference calls this path from six tf.data workers (utils/datasets.py:42)
    import torch
    cfg = ssd_config(128)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    errors = []

    def worker(seed):
        try:
            torch.cuda.set_device(0)
            rng = np.random.default_rng(seed)
            boxes, labels, counts = S.make_ground_truth(rng, 2, 128, gmax=16)
            y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=8)
            for _ in range(5):
                conf, locs = fda.match_encode_batch(grid, boxes, counts, labels, 128.0)
                dets, keep, count = fda.decode_nms_batch(grid, y, 0.05)
            for b in range(2):
                c_conf, c_locs = C.compute_target(anchors_np, boxes[b, :counts[b]],
                                                  labels[b, :counts[b]], 128.0)
                np.testing.assert_array_equal(conf[b], c_conf)
                assert_bits_equal(locs[b], c_locs)
                c_dets, c_idx = C.postprocess(y[b], anchors_np, 0.05)
                np.testing.assert_array_equal(keep[b, :count[b]], c_idx)
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(100 + i,)) for i in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_context_on_another_stream_is_ordered(fda):
    """A caller-supplied context bound to a stream that is not the current one: inputs produced on
    the current stream are seen, results are complete when the current stream (or the host) reads
    them."""
    import torch
    from face_detection_amd.runtime import Context
    cfg = ssd_config(256)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    rng = np.random.default_rng(31)
    boxes, labels, counts = S.make_ground_truth(rng, 6, 256, gmax=32)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=8)
    side = torch.cuda.Stream()
    ctx = Context(0, side)
    assert side.cuda_stream != torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        # inputs written on the current stream right before the call
        d_boxes = torch.from_numpy(boxes).cuda() * 1.0
        d_labels = torch.from_numpy(labels).cuda() * 1.0
        d_y = torch.from_numpy(y).cuda() * 1.0
        conf, locs = fda.match_encode_batch(grid, d_boxes, counts, d_labels, 256.0, context=ctx)
        dets, keep, count = fda.decode_nms_batch(grid, d_y, 0.05, context=ctx)
        total = conf.sum() + locs.sum() + dets.sum()            # consumed on the current stream
        n_conf, n_locs = fda.match_encode_batch(grid, boxes, counts, labels, 256.0, context=ctx)   # numpy out
        assert torch.isfinite(total)
        for b in range(6):
            g = counts[b]
            c_conf, c_locs = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g], 256.0)
            np.testing.assert_array_equal(conf[b].cpu().numpy(), c_conf)
            np.testing.assert_array_equal(n_conf[b], c_conf)
            assert_bits_equal(locs[b].cpu().numpy(), c_locs)
            assert_bits_equal(n_locs[b], c_locs)
            c_dets, c_idx = C.postprocess(y[b], anchors_np, 0.05)
            n = int(count[b])
            np.testing.assert_array_equal(keep[b, :n].cpu().numpy(), c_idx)
            assert_bits_equal(dets[b, :n].cpu().numpy(), c_dets)
    # a raw stream handle: numpy in and out, the arrays live in library memory on that stream
    r_dets, r_keep, r_count = fda.decode_nms_batch(grid, y, 0.05, context=Context(0, side.cuda_stream))
    assert isinstance(r_dets, np.ndarray)
    np.testing.assert_array_equal(r_count, count.cpu().numpy())
    np.testing.assert_array_equal(r_keep, keep.cpu().numpy())
    assert_bits_equal(r_dets, dets.cpu().numpy())
    with pytest.raises(ValueError):                            # torch tensors need a torch stream
        fda.decode_nms_batch(grid, torch.from_numpy(y).cuda(), 0.05, context=Context(0, side.cuda_stream))


def test_context_set_stream(fda):
    """ssdbox_ctx_set_stream: from a caller's stream to a stream of the context's own and back."""
    import ctypes
    import torch
    from face_detection_amd import _lib
    from face_detection_amd.runtime import Context
    lib = _lib.load()
    cfg = ssd_config(128)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    rng = np.random.default_rng(32)
    boxes, labels, counts = S.make_ground_truth(rng, 2, 128, gmax=8)
    ctx = Context(0)
    want = [C.compute_target(anchors_np, boxes[b, :counts[b]], labels[b, :counts[b]], 128.0) for b in range(2)]
    d = [torch.from_numpy(x).cuda() for x in (boxes, labels, counts)]
    A = len(anchors_np)

    def run():
        conf = torch.empty((2, A), dtype=torch.float32, device="cuda")
        locs = torch.empty((2, A, 4), dtype=torch.float32, device="cuda")
        _lib.check(lib.ssdbox_match_encode(ctx.handle, grid.geometry.handle, None, A, *(ctypes.c_void_p(t.data_ptr()) for t in d),
                                           2, 8, 128.0, 0.5, ctypes.c_void_p(conf.data_ptr()),
                                           ctypes.c_void_p(locs.data_ptr()), None, None))
        ctx.synchronize()
        for b in range(2):
            np.testing.assert_array_equal(conf[b].cpu().numpy(), want[b][0])
            assert_bits_equal(locs[b].cpu().numpy(), want[b][1])

    torch.cuda.synchronize()
    run()
    _lib.check(lib.ssdbox_ctx_set_stream(ctx.handle, None, 1))                       # a stream of its own
    run()
    _lib.check(lib.ssdbox_ctx_set_stream(ctx.handle, None, 1))                       # own -> own
    run()
    s = torch.cuda.Stream()
    _lib.check(lib.ssdbox_ctx_set_stream(ctx.handle, ctypes.c_void_p(s.cuda_stream), 0))   # own -> caller's
    run()
