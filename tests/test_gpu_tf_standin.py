"""The tf.Tensor branch of the host layer, against a STAND-IN for TensorFlow.

TensorFlow is not installed where this repository is built, so its tensors have never gone
through these calls.  What runs here is the code a tf.Tensor takes (runtime.Session.put / get:
tf.cast, tf.experimental.dlpack.to_dlpack / from_dlpack, .numpy(), tf.convert_to_tensor), driven
by a module named `tensorflow` that offers only that surface.  Its GPU tensors are torch CUDA
tensors underneath (torch plays TensorFlow's allocator; the library itself still runs without
torch: no argument is a torch tensor).  This pins the branch's logic, not TensorFlow's behaviour."""
import sys
import types

import numpy as np
import pytest

from conftest import ssd_config
from oracle import c_oracle as C
from face_detection_amd import synthetic as S

pytestmark = pytest.mark.gpu
F32 = np.float32


class _DType:
    def __init__(self, np_dtype):
        self.as_numpy_dtype = np.dtype(np_dtype).type

    def __eq__(self, other):
        return np.dtype(self.as_numpy_dtype) == np.dtype(getattr(other, "as_numpy_dtype", other))

    def __hash__(self):
        return hash(np.dtype(self.as_numpy_dtype))


def _make_tf(gpu=True, dlpack_in=True, dlpack_out=True):
    """A module with the few names the host layer uses.  `gpu`: tensors live on the device;
    `dlpack_in` / `dlpack_out`: whether to_dlpack / from_dlpack work (a TensorFlow build without a
    ROCm DLPack producer raises there, and the host layer must fall back to host copies)."""
    import torch
    tf = types.ModuleType("tensorflow")
    calls = {"to_dlpack": 0, "from_dlpack": 0, "numpy": 0, "convert": 0}

    class EagerTensor:
        __module__ = "tensorflow.python.framework.ops"

        def __init__(self, t):
            self._t = t                                   # torch tensor, cuda or cpu

        @property
        def shape(self):
            return tuple(self._t.shape)

        @property
        def dtype(self):
            return _DType(self._t.numpy().dtype if not self._t.is_cuda else self._t.cpu().numpy().dtype)

        def numpy(self):
            calls["numpy"] += 1
            return self._t.cpu().numpy()

        def __array__(self, dtype=None, copy=None):
            return self.numpy() if dtype is None else self.numpy().astype(dtype)

        def __getitem__(self, key):
            return EagerTensor(self._t[key])

    def convert_to_tensor(a):
        calls["convert"] += 1
        t = torch.from_numpy(np.ascontiguousarray(a))
        return EagerTensor(t.cuda() if gpu else t)

    def cast(x, dtype):
        return EagerTensor(x._t.to({np.dtype(np.float32): torch.float32,
                                    np.dtype(np.int32): torch.int32}[np.dtype(dtype)]))

    def to_dlpack(x):
        calls["to_dlpack"] += 1
        if not dlpack_in:
            raise RuntimeError("stand-in: no DLPack producer for this device")
        return torch.utils.dlpack.to_dlpack(x._t)

    def from_dlpack(capsule):
        calls["from_dlpack"] += 1
        if not dlpack_out:
            raise RuntimeError("stand-in: no DLPack consumer for this device")   # capsule left unconsumed
        return EagerTensor(torch.utils.dlpack.from_dlpack(capsule))

    tf.convert_to_tensor, tf.cast, tf.is_tensor = convert_to_tensor, cast, lambda x: isinstance(x, EagerTensor)
    tf.experimental = types.ModuleType("tensorflow.experimental")
    tf.experimental.dlpack = types.ModuleType("tensorflow.experimental.dlpack")
    tf.experimental.dlpack.to_dlpack, tf.experimental.dlpack.from_dlpack = to_dlpack, from_dlpack
    tf.EagerTensor, tf.calls = EagerTensor, calls
    return tf


@pytest.fixture(params=[(True, True, True), (True, False, False), (False, True, True), (True, True, False)],
                ids=["gpu-dlpack", "gpu-no-dlpack", "host", "gpu-export-refused"])
def tf(request, monkeypatch):
    mod = _make_tf(*request.param)
    monkeypatch.setitem(sys.modules, "tensorflow", mod)
    monkeypatch.setitem(sys.modules, "tensorflow.experimental", mod.experimental)
    monkeypatch.setitem(sys.modules, "tensorflow.experimental.dlpack", mod.experimental.dlpack)
    mod.params = request.param
    return mod


def bits(a):
    return np.ascontiguousarray(a, F32).view(np.uint32)


def test_drop_in_calls_take_and_return_tf_tensors(tf):
    from face_detection_amd.utils import anchor_utils, box_utils
    gpu, dl_in, dl_out = tf.params
    cfg = ssd_config(256)
    grid = anchor_utils.generate_anchors(cfg)
    anchors = C.generate_anchors(cfg)
    rng = np.random.default_rng(11)
    boxes, labels, counts = S.make_ground_truth(rng, 1, 256, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=16)
    g = int(counts[0])
    gt, lab = np.ascontiguousarray(boxes[0, :g]), np.ascontiguousarray(labels[0, :g])

    conf, locs = box_utils.compute_target(grid, tf.convert_to_tensor(gt), tf.convert_to_tensor(lab), 256.0)
    assert tf.is_tensor(conf) and tf.is_tensor(locs)
    want_conf, want_locs = C.compute_target(anchors, gt, lab, 256.0)
    np.testing.assert_array_equal(conf.numpy(), want_conf)
    np.testing.assert_array_equal(bits(locs.numpy()), bits(want_locs))

    iou = box_utils.compute_iou(tf.convert_to_tensor(gt), tf.convert_to_tensor(gt))
    assert tf.is_tensor(iou)
    np.testing.assert_array_equal(bits(iou.numpy()), bits(C.compute_iou(gt, gt)))

    dec = box_utils.decode(grid, tf.convert_to_tensor(np.ascontiguousarray(y[0, :, 2:])))
    want_dec = C.decode(anchors, y[0, :, 2:])
    np.testing.assert_array_equal(bits(dec.numpy()), bits(want_dec))

    mask = y[0, :, 1] > 0.05
    cand, scores = np.ascontiguousarray(want_dec[mask]), np.ascontiguousarray(y[0, mask, 1])
    keep = box_utils.compute_nms(tf.convert_to_tensor(cand), tf.convert_to_tensor(scores), 0.45, 200)
    assert tf.is_tensor(keep)
    np.testing.assert_array_equal(keep.numpy(), C.compute_nms(cand, scores, 0.45, 200))

    # which road the tensors took
    if dl_in:
        assert tf.calls["to_dlpack"] > 0
    else:
        assert tf.calls["numpy"] > 0                       # host copies instead
    assert tf.calls["from_dlpack"] > 0
    if not dl_out:
        assert tf.calls["convert"] > 8                     # results came back through convert_to_tensor


def test_batched_calls_take_and_return_tf_tensors(tf):
    import face_detection_amd as fda
    from face_detection_amd.utils import anchor_utils
    cfg = ssd_config(256)
    grid = anchor_utils.generate_anchors(cfg)
    anchors = C.generate_anchors(cfg)
    A = anchors.shape[0]
    rng = np.random.default_rng(5)
    B = 3
    boxes, labels, counts = S.make_ground_truth(rng, B, 256, gmax=24)
    y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=16)
    y_true = fda.match_encode_batch(grid, tf.convert_to_tensor(boxes), tf.convert_to_tensor(counts),
                                    tf.convert_to_tensor(labels), 256.0, y_true=True)
    assert tf.is_tensor(y_true) and y_true.shape == (B, A, 6)
    got = y_true.numpy()
    for b in range(B):
        g = int(counts[b])
        conf, locs = C.compute_target(anchors, boxes[b, :g], labels[b, :g], 256.0)
        np.testing.assert_array_equal(bits(got[b]), bits(C.y_true(conf, locs)))

    dets, keep_idx, count = fda.decode_nms_batch(grid, tf.convert_to_tensor(y), 0.05, 0.45, 1000)
    assert tf.is_tensor(dets) and tf.is_tensor(keep_idx) and tf.is_tensor(count)
    dets, keep_idx, count = dets.numpy(), keep_idx.numpy(), count.numpy()
    for b in range(B):
        c_dets, c_keep = C.postprocess(y[b], anchors, 0.05, 1.0, 1.0, 0.45, 1000)
        n = int(count[b])
        assert n == len(c_keep)
        np.testing.assert_array_equal(keep_idx[b, :n], c_keep)
        np.testing.assert_array_equal(bits(dets[b, :n]), bits(c_dets))
