"""The benchmark configurations that the other GPU tests do not reach, against the C oracle:
decode+NMS at 1024x1024 (batch 16, both anchor sets, more than 8192 candidates per image at
threshold 0), the batch of 256 images @640x640 in one call against the eight per-GPU shards of
32, a single 640x640 image, and a two-rank rehearsal of bench.py on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ssd_config
from oracle import box_oracle as O
from oracle import c_oracle as C
from face_detection_amd import synthetic as S
from face_detection_amd.dist import shard_range

pytestmark = pytest.mark.gpu
F32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fda():
    import face_detection_amd as pkg
    from face_detection_amd.utils import anchor_utils, box_utils
    pkg.anchor_utils = anchor_utils
    pkg.box_utils = box_utils
    return pkg


def bits(a):
    return np.ascontiguousarray(a, F32).view(np.uint32)


@pytest.mark.parametrize("ratios,n_anchors", [("config", 261120), ("model", 348160)])
def test_decode_nms_1024_batch16(fda, ratios, n_anchors):
    rng = np.random.default_rng(64)
    cfg = ssd_config(1024, ratios)
    anchors_np = O.generate_anchors(cfg)
    assert len(anchors_np) == n_anchors
    grid = fda.anchor_utils.generate_anchors(cfg)
    boxes, _, counts = S.make_ground_truth(rng, 16, 1024, gmax=256)
    y = S.make_head_outputs(rng, cfg, boxes, counts)
    wh = rng.uniform(300, 3000, (16, 2)).astype(F32)
    for thr in (0.05, 0.0):
        dets, keep, count = fda.decode_nms_batch(grid, y, thr, 0.45, 1000, wh)
        n_cand = (y[:, :, 1] > thr).sum(1)
        assert n_cand.min() > (8192 if thr == 0.0 else 1000)
        for b in (0, 7, 15, int(np.argmax(counts))):           # bit-exact against the oracle
            c_dets, c_idx = C.postprocess(y[b], anchors_np, thr, float(wh[b, 0]), float(wh[b, 1]), 0.45, 1000)
            n = int(count[b])
            assert n == len(c_idx)
            np.testing.assert_array_equal(keep[b, :n], c_idx)
            np.testing.assert_array_equal(bits(dets[b, :n]), bits(c_dets))
            assert (keep[b, n:] == -1).all() and (dets[b, n:] == 0).all()
        for b in range(16):                                     # properties that hold at any size
            n = int(count[b])
            assert 0 < n <= 1000
            assert (np.diff(dets[b, :n, 1]) <= 0).all()         # kept order is score order
            assert (y[b, keep[b, :n], 1] > thr).all()
            raw = C.decode(anchors_np[keep[b, :n]], y[b, keep[b, :n], 2:])
            iou = O.compute_iou(raw, raw)
            np.fill_diagonal(iou, 0)
            assert (iou <= F32(0.45)).all()                     # survivors never overlap above the threshold


def test_batch256_equals_eight_shards_of_32(fda):
    """Configuration 5 on one GPU: 256 images in one call equal the eight shard_range slices a
    node of eight GPUs would process, both halves, bit for bit; three images against the oracle."""
    rng = np.random.default_rng(65)
    cfg = ssd_config(640)
    grid = fda.anchor_utils.generate_anchors(cfg)
    anchors_np = O.generate_anchors(cfg)
    b64, l64, c64 = S.make_ground_truth(rng, 64, 640, gmax=256)
    y64 = S.make_head_outputs(rng, cfg, b64, c64)
    order = np.concatenate([rng.permutation(64) for _ in range(4)])       # 256 images, 64 distinct
    boxes, labels, counts = b64[order], l64[order], c64[order]
    import torch
    d_y = torch.from_numpy(y64).cuda()[torch.from_numpy(order).cuda()]    # (256, A, 6) on the device
    d_boxes, d_counts, d_labels = (torch.from_numpy(x).cuda() for x in (boxes, counts, labels))
    conf, locs = fda.match_encode_batch(grid, d_boxes, d_counts, d_labels, 640.0)
    dets, keep, count = fda.decode_nms_batch(grid, d_y, 0.05, 0.45, 1000)
    for r in range(8):
        lo, hi = shard_range(256, r, 8)
        assert hi - lo == 32
        s_conf, s_locs = fda.match_encode_batch(grid, d_boxes[lo:hi].contiguous(), d_counts[lo:hi].contiguous(),
                                                d_labels[lo:hi].contiguous(), 640.0)
        assert torch.equal(s_conf, conf[lo:hi]) and torch.equal(s_locs.view(torch.int32), locs[lo:hi].view(torch.int32))
        s_dets, s_keep, s_count = fda.decode_nms_batch(grid, d_y[lo:hi].contiguous(), 0.05, 0.45, 1000)
        assert torch.equal(s_count, count[lo:hi]) and torch.equal(s_keep, keep[lo:hi])
        assert torch.equal(s_dets.view(torch.int32), dets[lo:hi].view(torch.int32))
    for b in (0, 100, 255):
        g = int(counts[b])
        c_conf, c_locs = C.compute_target(anchors_np, boxes[b, :g], labels[b, :g], 640.0)
        np.testing.assert_array_equal(conf[b].cpu().numpy(), c_conf)
        np.testing.assert_array_equal(bits(locs[b].cpu().numpy()), bits(c_locs))
        c_dets, c_idx = C.postprocess(y64[order[b]], anchors_np, 0.05, 1.0, 1.0, 0.45, 1000)
        n = int(count[b])
        assert n == len(c_idx)
        np.testing.assert_array_equal(keep[b, :n].cpu().numpy(), c_idx)
        np.testing.assert_array_equal(bits(dets[b, :n].cpu().numpy()), bits(c_dets))


def test_single_image_640_equals_first_of_batch32(fda):
    """Configuration 2: one 640x640 head output (as reference inference.py processes images)
    gives what image 0 of a batch of 32 gives, and what the oracle gives."""
    rng = np.random.default_rng(66)
    for ratios in ("config", "model"):
        cfg = ssd_config(640, ratios)
        grid = fda.anchor_utils.generate_anchors(cfg)
        anchors_np = O.generate_anchors(cfg)
        boxes, _, counts = S.make_ground_truth(rng, 32, 640, gmax=256)
        y = S.make_head_outputs(rng, cfg, boxes, counts)
        wh = rng.uniform(300, 3000, (32, 2)).astype(F32)
        dets, keep, count = fda.decode_nms_batch(grid, y, 0.05, 0.45, 1000, wh)
        d1, k1, c1 = fda.decode_nms_batch(grid, y[:1], 0.05, 0.45, 1000, wh[:1])
        assert int(c1[0]) == int(count[0])
        np.testing.assert_array_equal(k1[0], keep[0])
        np.testing.assert_array_equal(bits(d1[0]), bits(dets[0]))
        c_dets, c_idx = C.postprocess(y[0], anchors_np, 0.05, float(wh[0, 0]), float(wh[0, 1]), 0.45, 1000)
        n = int(c1[0])
        assert n == len(c_idx)
        np.testing.assert_array_equal(k1[0, :n], c_idx)
        np.testing.assert_array_equal(bits(d1[0, :n]), bits(c_dets))


def test_bench_two_rank_rehearsal_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it, both ranks on this one GPU with gloo for the
    collective: one JSON line, the all-gather figure present and the gathered detections equal to
    what rank 0 computes itself from both ranks' inputs.  A rehearsal of the launch path, not a
    scaling measurement."""
    env = dict(os.environ, SSDBOX_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "8", "--warmup", "8", "--cpu-images", "2", "--seeds", "1", "--no-stress-leg"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["warmup"] == 8 and d["scaling"] == "weak"
    assert d["allgather_detections_ms"] > 0
    assert d["allgather_equals_local_recompute"] is True
    assert d["config"]["parallelism"] == "image-sharded x2"
    # lines of more than one rank carry the (reduced) CPU baseline and the parity gate too
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] == 1
    assert all(v == d["parity"]["images"] == 2 for v in d["parity"].values())
    assert d["roofline"]["frac"] > 0
