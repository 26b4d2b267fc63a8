"""Drop-in for the reference's utils/box_utils.py: same function names, argument order and
defaults (reference utils/box_utils.py:4,21,47,101,126,150,167,184), computed by libssdbox
HIP kernels on an MI355X.  Inputs may be numpy arrays, torch tensors, tf.Tensors or anything
exporting __dlpack__; results come back in the family of the first array argument.
There is no CPU path: without the built library or a GPU these functions raise.
"""
import numpy as np

from .. import _lib
from ..batched import _anchor_source, match_encode_batch
from ..runtime import AnchorGrid, Session, _is_torch_tensor, _ptr, _torch


def _rows4(t, what):
    if len(t.shape) < 1 or t.shape[-1] != 4:
        raise ValueError("%s must have a last dimension of 4, got %s" % (what, tuple(t.shape)))


def compute_area(top_left, bot_right, max_size=512.0):
    """Area of boxes given (..., 2) corners, sides clipped to [0, max_size]
    (reference utils/box_utils.py:4-18)."""
    sess = Session(top_left, bot_right)
    ctx = sess.ctx
    if _is_torch_tensor(top_left) or _is_torch_tensor(bot_right):
        torch = _torch()
        tl, family = sess.put(top_left, np.float32)
        br, _ = sess.put(bot_right, np.float32)
        tl, br = torch.broadcast_tensors(tl, br)
        tl, br = tl.contiguous(), br.contiguous()
    else:
        tl, family = sess.put(top_left, np.float32)
        br, fam_b = sess.put(bot_right, np.float32)
        if tuple(tl.shape) != tuple(br.shape):     # broadcast on the host, then upload
            h_tl, h_br = np.broadcast_arrays(sess.host(tl), sess.host(br))
            tl, br = sess.put(np.ascontiguousarray(h_tl), np.float32)[0], sess.put(np.ascontiguousarray(h_br), np.float32)[0]
    if tl.shape[-1] != 2:
        raise ValueError("corners must have a last dimension of 2")
    out = sess.empty(tuple(tl.shape[:-1]))
    _lib.check(ctx.lib.ssdbox_compute_area(ctx.handle, _ptr(tl), _ptr(br), out.numel(),
                                           float(max_size), _ptr(out)))
    return sess.get(out, family, top_left)


def compute_iou(boxes_a, boxes_b, max_size=512.0):
    """(Na,4),(Nb,4) corner boxes -> (Na,Nb) IoU, no epsilon
    (reference utils/box_utils.py:21-44)."""
    sess = Session(boxes_a, boxes_b)
    ctx = sess.ctx
    a, family = sess.put(boxes_a, np.float32)
    b, _ = sess.put(boxes_b, np.float32)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 4 or b.shape[1] != 4:
        raise ValueError("compute_iou takes (Na,4) and (Nb,4) boxes")
    out = sess.empty((a.shape[0], b.shape[0]))
    _lib.check(ctx.lib.ssdbox_compute_iou(ctx.handle, _ptr(a), a.shape[0], _ptr(b), b.shape[0],
                                          float(max_size), _ptr(out)))
    return sess.get(out, family, boxes_a)


def compute_target(default_boxes, gt_boxes, gt_labels, max_size=512.0, iou_threshold=0.5):
    """Classification and regression targets of one image
    (reference utils/box_utils.py:47-98): returns (gt_confs (A,), gt_locs (A,4)).
    default_boxes may be the object generate_anchors returned (anchors are then built on
    the fly) or any (A,4) [cx,cy,w,h] array."""
    gt = np.asarray(gt_boxes) if not hasattr(gt_boxes, "shape") else gt_boxes
    if len(gt.shape) != 2 or gt.shape[1] != 4:
        raise ValueError("gt_boxes must have shape (G, 4), got %s" % (tuple(gt.shape),))
    G = int(gt.shape[0])
    if G == 0:
        raise ValueError("compute_target needs at least one ground-truth box")
    sess = Session(default_boxes, gt_boxes, gt_labels)
    boxes, family = sess.put(gt_boxes, np.float32)
    labels, _ = sess.put(gt_labels, np.float32)
    if tuple(labels.shape) != (G,):
        raise ValueError("gt_labels must have shape (%d,)" % G)
    # (device arrays in, device arrays out: the batched call returns what it is given)
    conf, locs = match_encode_batch(default_boxes, boxes[None], np.array([G], np.int32),
                                    labels[None], max_size, iou_threshold, _device_results=True)
    conf, locs = conf[0], locs[0]
    label_dtype = getattr(gt_labels, "dtype", None)
    conf = sess.get(conf, family, gt_boxes)
    locs = sess.get(locs, family, gt_boxes)
    if family == "numpy" and label_dtype is not None:
        try:
            if np.dtype(label_dtype) != np.float32:
                conf = conf.astype(label_dtype)
        except TypeError:
            pass
    return conf, locs


def _codec(fn_name, default_boxes, values, variance):
    if variance is None:
        variance = [0.1, 0.2]
    sess = Session(default_boxes, values)
    ctx = sess.ctx
    v, family = sess.put(values, np.float32)
    _rows4(v, "boxes")
    geom, anc, A, _keep = _anchor_source(sess, default_boxes)
    if v.dim() < 2 or v.shape[-2] != A:
        raise ValueError("expected (..., %d, 4) values for %d default boxes, got %s"
                         % (A, A, tuple(v.shape)))
    out = sess.empty(tuple(v.shape))
    fn = getattr(ctx.lib, fn_name)
    _lib.check(fn(ctx.handle, geom, _ptr(anc), A, _ptr(v), v.numel() // 4, float(variance[0]),
                  float(variance[1]), _ptr(out)))
    return sess.get(out, family, values)


def encode(default_boxes, boxes, variance=None):
    """Regression offsets of corner `boxes` against [cx,cy,w,h] default boxes
    (reference utils/box_utils.py:101-123)."""
    return _codec("ssdbox_encode", default_boxes, boxes, variance)


def decode(default_boxes, locs, variance=None):
    """Corner boxes from offsets, broadcasting over leading batch dimensions of `locs`
    (reference utils/box_utils.py:126-147)."""
    return _codec("ssdbox_decode", default_boxes, locs, variance)


def _transform(fn_name, boxes):
    sess = Session(boxes)
    ctx = sess.ctx
    b, family = sess.put(boxes, np.float32)
    _rows4(b, "boxes")
    out = sess.empty(tuple(b.shape))
    _lib.check(getattr(ctx.lib, fn_name)(ctx.handle, _ptr(b), b.numel() // 4, _ptr(out)))
    return sess.get(out, family, boxes)


def transform_corner_to_center(boxes):
    """(xmin, ymin, xmax, ymax) -> (cx, cy, w, h) (reference utils/box_utils.py:150-164)."""
    return _transform("ssdbox_corner_to_center", boxes)


def transform_center_to_corner(boxes):
    """(cx, cy, w, h) -> (xmin, ymin, xmax, ymax) (reference utils/box_utils.py:167-181)."""
    return _transform("ssdbox_center_to_corner", boxes)


def compute_nms(boxes, scores, nms_threshold, limit=200):
    """Greedy NMS over the `limit` best-scoring boxes; returns the int32 indices of the kept
    boxes in descending-score order (reference utils/box_utils.py:184-222).  Empty input
    gives an empty int32 result."""
    sess = Session(boxes, scores)
    ctx = sess.ctx
    b, family = sess.put(boxes, np.float32)
    s, _ = sess.put(scores, np.float32)
    n = int(b.shape[0])
    if n == 0:
        return sess.get(sess.empty((0,), np.int32), family, boxes)
    if b.dim() != 2 or b.shape[1] != 4 or tuple(s.shape) != (n,):
        raise ValueError("compute_nms takes (N,4) boxes and (N,) scores")
    if not float(nms_threshold) < 1.0:
        raise ValueError("nms_threshold must be < 1 (the reference loop never ends otherwise)")
    k = min(int(limit), n)
    keep = sess.empty((k,), np.int32)
    count = sess.empty((1,), np.int32)
    _lib.check(ctx.lib.ssdbox_compute_nms(ctx.handle, _ptr(b), _ptr(s), n, float(nms_threshold),
                                          int(limit), _ptr(keep), _ptr(count)))
    keep = keep[: int(sess.host(count)[0])]
    return sess.get(keep, family, boxes)
