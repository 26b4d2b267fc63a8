"""Randomised differential test of the HIP path against the C oracle: geometries, ground truth
and head outputs no hand-written case covers (non-square images, one to four layers, any
mix of aspect ratios, degenerate and out-of-range boxes, NaN / infinite / tied scores, offsets
that overflow exp).  Same bar as the parity tests: indices and masks equal, floats bit for bit.
"""
import os

import numpy as np
import pytest

from oracle import box_oracle as O
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu
F32 = np.float32
SEEDS = range(int(os.environ.get("SSDBOX_FUZZ_SEEDS", "10")))      # more for a longer hunt


@pytest.fixture(scope="module")
def fda():
    import face_detection_amd as pkg
    from face_detection_amd.utils import anchor_utils
    pkg.anchor_utils = anchor_utils
    return pkg


def random_config(rng):
    w, h = int(rng.integers(48, 321)), int(rng.integers(48, 321))
    n = int(rng.integers(1, 5))
    steps = sorted(rng.choice([4, 8, 16, 32], size=n, replace=False).tolist())
    steps = [s for s in steps if min(w, h) // s >= 1] or [4]
    n = len(steps)
    scales = np.sort(rng.uniform(0.01, 0.9, n + 1)).tolist()
    pool = [0.5, 0.9, 1.0, 1.8, 2.0, 2.5, 3.0]
    ratios = [sorted(rng.choice(pool, size=int(rng.integers(1, 5)), replace=False).tolist()) for _ in range(n)]
    return {'img_height': h, 'img_width': w, 'image_size': (h, w), 'offsets': float(rng.choice([0.5, 0.25, 0.0])),
            'n_classes': 1, 'scales': scales, 'aspect_ratios_per_layer': ratios,
            'two_boxes_for_ar1': bool(rng.integers(0, 2)), 'steps': steps, 'variances': [0.1, 0.2]}


def random_ground_truth(rng, B, gmax):
    counts = rng.integers(1, gmax + 1, B).astype(np.int32)
    boxes = np.zeros((B, gmax, 4), F32)
    labels = np.zeros((B, gmax), F32)
    for b in range(B):
        g = counts[b]
        c = rng.uniform(-0.05, 1.05, (g, 2))
        wh = np.exp(rng.uniform(np.log(0.01), np.log(0.8), (g, 2)))
        bx = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(F32)
        kind = rng.integers(0, 12, g)
        bx[kind == 0, 2] = bx[kind == 0, 0]                    # zero width
        bx[kind == 1, 3] = bx[kind == 1, 1] - F32(0.01)        # negative height
        bx[kind == 2] = bx[rng.integers(0, g)]                 # exact duplicate of another box
        boxes[b, :g] = bx
        labels[b, :g] = np.where(rng.uniform(size=g) < 0.9, 1.0, 0.0)
    return boxes, labels, counts


def bits(a):
    return np.ascontiguousarray(a, F32).view(np.uint32)


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_match_encode(fda, seed):
    rng = np.random.default_rng(7000 + seed)
    cfg = random_config(rng)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    np.testing.assert_array_equal(bits(np.asarray(grid)), bits(anchors_np))
    boxes, labels, counts = random_ground_truth(rng, 3, int(rng.integers(1, 60)))
    max_size = float(rng.choice([512.0, float(max(cfg['image_size'])), 0.5, 0.2]))
    thr = float(rng.choice([0.5, 0.3, 0.7, 0.05]))
    for anchors in (grid, anchors_np):
        conf, locs, midx, miou = fda.match_encode_batch(anchors, boxes, counts, labels, max_size, thr,
                                                        return_match=True)
        ytrue = fda.match_encode_batch(anchors, boxes, counts, labels, max_size, thr, y_true=True)
        for b in range(3):
            g = counts[b]
            c_conf, c_locs, c_idx, c_iou, _ = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g],
                                                               max_size, thr, details=True)
            np.testing.assert_array_equal(midx[b], c_idx, err_msg=repr(cfg))
            np.testing.assert_array_equal(bits(miou[b]), bits(c_iou), err_msg=repr(cfg))
            np.testing.assert_array_equal(conf[b], c_conf, err_msg=repr(cfg))
            np.testing.assert_array_equal(bits(locs[b]), bits(c_locs), err_msg=repr(cfg))
            np.testing.assert_array_equal(bits(ytrue[b]), bits(C.y_true(c_conf, c_locs)), err_msg=repr(cfg))


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_decode_nms(fda, seed):
    rng = np.random.default_rng(8000 + seed)
    cfg = random_config(rng)
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    A, B = len(anchors_np), 3
    y = np.empty((B, A, 6), F32)
    p = rng.beta(0.3, 3.0, (B, A)).astype(F32)
    y[..., 1] = p
    y[..., 0] = F32(1.0) - p
    y[..., 2:] = rng.normal(0, 1.5, (B, A, 4)).astype(F32)
    n_odd = max(A // 50, 1)
    for b in range(B):
        idx = rng.choice(A, size=min(6 * n_odd, A), replace=False)
        parts = np.array_split(idx, 6)
        y[b, parts[0], 1] = np.nan
        y[b, parts[1], 1] = np.inf
        y[b, parts[2], 1] = F32(0.75)                         # a tie group
        y[b, parts[3], 1] = -0.0
        y[b, parts[4], 4] = 500.0                             # exp overflows: infinite width
        y[b, parts[5], 2] = np.nan                            # NaN box
    thr = float(rng.choice([0.05, 0.3, 0.0, -1.0, 0.74]))
    nms_thr = float(rng.choice([0.45, 0.0, 0.9, -0.5]))
    limit = int(rng.choice([1000, 200, 33, 1, 1500, 4500, 9000], p=[0.4, 0.2, 0.1, 0.1, 0.08, 0.07, 0.05]))
    wh = rng.uniform(50, 3000, (B, 2)).astype(F32)
    for anchors in (grid, anchors_np):
        dets, keep, count = fda.decode_nms_batch(anchors, y, thr, nms_thr, limit, wh)
        for b in range(B):
            c_dets, c_idx = C.postprocess(y[b], anchors_np, thr, float(wh[b, 0]), float(wh[b, 1]), nms_thr, limit)
            n = int(count[b])
            assert n == len(c_idx), (n, len(c_idx), cfg, thr, nms_thr, limit)
            np.testing.assert_array_equal(keep[b, :n], c_idx, err_msg=repr((cfg, thr, nms_thr, limit)))
            np.testing.assert_array_equal(bits(dets[b, :n]), bits(c_dets), err_msg=repr((cfg, thr, nms_thr, limit)))
            assert (keep[b, n:] == -1).all() and (dets[b, n:] == 0).all()


@pytest.mark.parametrize("seed", SEEDS)
def test_fuzz_ssd_loss(fda, seed):
    rng = np.random.default_rng(9000 + seed)
    B, A = int(rng.integers(1, 7)), int(rng.integers(1, 5000))
    pos = (rng.uniform(size=(B, A)) < rng.choice([0.0, 0.01, 0.2])).astype(F32)
    yt = np.zeros((B, A, 6), F32)
    yt[..., 1] = pos
    yt[..., 0] = 1 - pos
    yt[..., 2:] = (rng.normal(0, 1.5, (B, A, 4)) * pos[..., None]).astype(F32)
    yp = np.empty((B, A, 6), F32)
    p1 = rng.beta(0.5, 4.0, (B, A)).astype(F32)
    if seed % 2:
        p1 = np.round(p1 * 16) / 16                            # many equal losses
    yp[..., 1] = p1
    yp[..., 0] = F32(1.0) - p1
    yp[..., 2:] = rng.normal(0, 1.5, (B, A, 4)).astype(F32)
    flat = yp.reshape(-1, 6)
    odd = rng.choice(len(flat), size=max(len(flat) // 40, 1), replace=False)
    flat[odd[::3], 0] = 0.0                                    # clipped at 1e-15
    flat[odd[1::3], 0] = 1e-20
    flat[odd[2::3], 1] = -0.5
    if seed % 3 == 0:
        flat[odd[::5], 0] = np.nan
    kw = dict(neg_pos_ratio=int(rng.integers(0, 6)), n_neg_min=int(rng.choice([0, 0, 7, 100])),
              alpha=float(rng.choice([1.0, 0.5, 2.0])))
    total, keep, grad = fda.ssd_loss_batch(yt, yp, return_keep=True, return_grad=True, **kw)
    c_total, c_keep, _, _, c_grad = C.ssd_loss(yt, yp, grad=True, **kw)
    np.testing.assert_array_equal(keep, c_keep, err_msg=repr((B, A, kw)))
    np.testing.assert_allclose(total, c_total, rtol=1e-6, equal_nan=True, err_msg=repr((B, A, kw)))
    np.testing.assert_array_equal(bits(grad), bits(c_grad), err_msg=repr((B, A, kw)))


@pytest.mark.parametrize("gmax", [256, 300])
def test_match_encode_widest_cells_and_tallest_tiles(fda, gmax):
    """Fifteen box shapes per cell (the most the geometry takes) on a 640-row image: the largest
    per-tile tables the match kernel builds, with one full list of boxes (256) and with more than
    one list (300: the chunked walk)."""
    rng = np.random.default_rng(99)
    ratios = [round(0.4 + 0.2 * i, 2) for i in range(14)] + [1.0]          # 14 + the 1.0 pair = 16 > limit
    ratios = sorted(set(ratios))[:14]                                       # 13 plain + 1.0 (two boxes) = 15
    assert 1.0 in ratios
    cfg = {'img_height': 640, 'img_width': 96, 'image_size': (640, 96), 'offsets': 0.5, 'n_classes': 1,
           'scales': [0.05, 0.2, 0.5], 'aspect_ratios_per_layer': [ratios, ratios[:3]],
           'two_boxes_for_ar1': True, 'steps': [4, 32], 'variances': [0.1, 0.2]}
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    np.testing.assert_array_equal(bits(np.asarray(grid)), bits(anchors_np))
    boxes, labels, counts = random_ground_truth(rng, 2, gmax)
    counts[0] = gmax
    boxes0, labels0, _ = random_ground_truth(rng, 1, gmax)
    boxes[0], labels[0] = boxes0[0], labels0[0]
    conf, locs, midx, miou = fda.match_encode_batch(grid, boxes, counts, labels, 640.0, 0.5, return_match=True)
    for b in range(2):
        g = counts[b]
        c_conf, c_locs, c_idx, c_iou, _ = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g], 640.0, 0.5,
                                                           details=True)
        np.testing.assert_array_equal(np.asarray(midx[b]), c_idx)
        np.testing.assert_array_equal(bits(miou[b]), bits(c_iou))
        np.testing.assert_array_equal(bits(conf[b]), bits(c_conf))
        np.testing.assert_array_equal(bits(locs[b]), bits(c_locs))


CROWDED_SEEDS = range(int(os.environ.get("SSDBOX_FUZZ_CROWDED_SEEDS", "4")))


@pytest.mark.parametrize("seed", CROWDED_SEEDS)
def test_fuzz_match_encode_crowded(fda, seed):
    """Random geometries on larger images with full and overfull box lists: tall tiles, long
    walks, more than one list of 256 boxes per image."""
    rng = np.random.default_rng(9000 + seed)
    cfg = random_config(rng)
    cfg['img_height'], cfg['img_width'] = int(rng.integers(320, 700)), int(rng.integers(96, 400))
    cfg['image_size'] = (cfg['img_height'], cfg['img_width'])
    anchors_np = O.generate_anchors(cfg)
    grid = fda.anchor_utils.generate_anchors(cfg)
    gmax = int(rng.choice([200, 256, 257, 300, 600]))
    boxes, labels, counts = random_ground_truth(rng, 3, gmax)
    counts[0] = gmax
    boxes0, labels0, _ = random_ground_truth(rng, 1, gmax)
    boxes[0, :], labels[0, :] = boxes0[0], labels0[0]
    max_size = float(max(cfg['image_size']))
    conf, locs, midx, miou = fda.match_encode_batch(grid, boxes, counts, labels, max_size, 0.5, return_match=True)
    for b in range(3):
        g = counts[b]
        c_conf, c_locs, c_idx, c_iou, _ = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g], max_size, 0.5,
                                                           details=True)
        np.testing.assert_array_equal(np.asarray(midx[b]), c_idx)
        np.testing.assert_array_equal(bits(miou[b]), bits(c_iou))
        np.testing.assert_array_equal(bits(conf[b]), bits(c_conf))
        np.testing.assert_array_equal(bits(locs[b]), bits(c_locs))
