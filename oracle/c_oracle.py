"""ctypes binding of the scalar C oracle (oracle/ssdbox_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Same conventions as oracle/box_oracle.py (parity unpinned by reference tests); exp/log are
the fixed Cephes float32 sequences, so results can be compared bit for bit with another
implementation of the same sequences.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libssdbox_oracle.so")
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i = ctypes.POINTER(ctypes.c_int32)
c_d = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    src = os.path.join(_HERE, "ssdbox_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libssdbox_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()                      # also rebuilds a library older than its source
        L = ctypes.CDLL(_SO)
        L.orc_logf.restype = ctypes.c_float
        L.orc_logf.argtypes = [ctypes.c_float]
        L.orc_expf.restype = ctypes.c_float
        L.orc_expf.argtypes = [ctypes.c_float]
        L.orc_generate_anchors.restype = ctypes.c_int64
        L.orc_generate_anchors.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_i, c_d,
                                           c_i, c_d, ctypes.c_int, ctypes.c_double, c_f]
        L.orc_compute_iou.restype = None
        L.orc_compute_iou.argtypes = [c_f, ctypes.c_int64, c_f, ctypes.c_int64, ctypes.c_float, c_f]
        L.orc_encode.restype = None
        L.orc_encode.argtypes = [c_f, ctypes.c_int64, c_f, ctypes.c_int64, c_f]
        L.orc_decode.restype = None
        L.orc_decode.argtypes = [c_f, ctypes.c_int64, c_f, ctypes.c_int64, c_f]
        L.orc_compute_target.restype = ctypes.c_int
        L.orc_compute_target.argtypes = [c_f, ctypes.c_int64, c_f, c_f, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_float, c_f, c_f, c_i, c_f, c_i]
        L.orc_y_true.restype = None
        L.orc_y_true.argtypes = [c_f, c_f, ctypes.c_int64, c_f]
        L.orc_nms.restype = ctypes.c_int
        L.orc_nms.argtypes = [c_f, c_f, ctypes.c_int64, ctypes.c_float, ctypes.c_int, c_i]
        L.orc_postprocess.restype = ctypes.c_int
        L.orc_postprocess.argtypes = [c_f, c_f, ctypes.c_int64, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_int, ctypes.c_float, ctypes.c_float, c_f, c_i]
        L.orc_softmax2.restype = None
        L.orc_softmax2.argtypes = [c_f, ctypes.c_int64, c_f]
        L.orc_ssd_loss.restype = ctypes.c_int
        L.orc_ssd_loss.argtypes = [c_f, c_f, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_float, c_f, ctypes.POINTER(ctypes.c_ubyte), c_f, c_f,
                                   ctypes.POINTER(ctypes.c_double)]
        _lib = L
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(c_f)


def logf(x):
    L = lib()
    return np.array([L.orc_logf(float(v)) for v in np.asarray(x, np.float32).ravel()],
                    np.float32).reshape(np.shape(x))


def expf(x):
    L = lib()
    return np.array([L.orc_expf(float(v)) for v in np.asarray(x, np.float32).ravel()],
                    np.float32).reshape(np.shape(x))


def generate_anchors(config):
    steps = np.asarray(config['steps'], np.int32)
    scales = np.asarray(config['scales'], np.float64)
    ratios = config['aspect_ratios_per_layer']
    counts = np.asarray([len(r) for r in ratios], np.int32)
    flat = np.asarray([x for r in ratios for x in r], np.float64)
    args = (int(config['img_width']), int(config['img_height']), len(steps),
            steps.ctypes.data_as(c_i), scales.ctypes.data_as(c_d), counts.ctypes.data_as(c_i),
            flat.ctypes.data_as(c_d), int(bool(config['two_boxes_for_ar1'])),
            float(config['offsets']))
    n = lib().orc_generate_anchors(*args, None)
    out = np.empty((n, 4), np.float32)
    lib().orc_generate_anchors(*args, out.ctypes.data_as(c_f))
    return out


def compute_iou(a, b, max_size=512.0):
    a, pa = _f(a)
    b, pb = _f(b)
    out = np.empty((len(a), len(b)), np.float32)
    lib().orc_compute_iou(pa, len(a), pb, len(b), max_size, out.ctypes.data_as(c_f))
    return out


def encode(anchors, boxes):
    anchors, pa = _f(anchors)
    boxes, pb = _f(boxes)
    out = np.empty_like(boxes)
    lib().orc_encode(pa, len(anchors), pb, boxes.size // 4, out.ctypes.data_as(c_f))
    return out


def decode(anchors, locs):
    anchors, pa = _f(anchors)
    locs, pl = _f(locs)
    out = np.empty_like(locs)
    lib().orc_decode(pa, len(anchors), pl, locs.size // 4, out.ctypes.data_as(c_f))
    return out


def compute_target(anchors, gt_boxes, gt_labels, max_size=512.0, iou_threshold=0.5,
                   details=False):
    anchors, pa = _f(anchors)
    gt, pg = _f(gt_boxes)
    lab, pl = _f(gt_labels)
    A, G = len(anchors), len(gt)
    conf = np.empty((A,), np.float32)
    locs = np.empty((A, 4), np.float32)
    idx = np.empty((A,), np.int32)
    iou = np.empty((A,), np.float32)
    ba = np.empty((G,), np.int32)
    rc = lib().orc_compute_target(pa, A, pg, pl, G, max_size, iou_threshold,
                                  conf.ctypes.data_as(c_f), locs.ctypes.data_as(c_f),
                                  idx.ctypes.data_as(c_i), iou.ctypes.data_as(c_f),
                                  ba.ctypes.data_as(c_i))
    if rc != 0:
        raise ValueError("compute_target needs at least one anchor and one GT box")
    if details:
        return conf, locs, idx, iou, ba
    return conf, locs


def y_true(conf, locs):
    conf, pc = _f(conf)
    locs, pl = _f(locs)
    y = np.empty((len(conf), 6), np.float32)
    lib().orc_y_true(pc, pl, len(conf), y.ctypes.data_as(c_f))
    return y


def compute_nms(boxes, scores, nms_threshold, limit=200):
    boxes, pb = _f(boxes)
    scores, ps = _f(scores)
    n = len(scores)
    keep = np.empty((max(min(n, limit), 1),), np.int32)
    k = lib().orc_nms(pb, ps, n, nms_threshold, limit, keep.ctypes.data_as(c_i))
    return keep[:k].copy()


def postprocess(y_pred_image, anchors, conf_thresh, img_width=1.0, img_height=1.0,
                nms_threshold=0.45, limit=1000):
    y, py = _f(y_pred_image)
    anchors, pa = _f(anchors)
    dets = np.zeros((limit, 6), np.float32)
    idx = np.zeros((limit,), np.int32)
    k = lib().orc_postprocess(py, pa, len(anchors), conf_thresh, nms_threshold, limit,
                              img_width, img_height, dets.ctypes.data_as(c_f),
                              idx.ctypes.data_as(c_i))
    return dets[:k].copy(), idx[:k].copy()


def softmax2(y_logits):
    """(..., 6) head rows with raw class logits -> the same rows with softmaxed class columns."""
    y, py = _f(y_logits)
    out = np.empty_like(y)
    lib().orc_softmax2(py, y.size // 6, out.ctypes.data_as(c_f))
    return out


def ssd_loss(y_true, y_pred, neg_pos_ratio=3, n_neg_min=0, alpha=1.0, grad=False):
    """(total (B,), keep (B,A) bool, cls (B,A), n_positive[, grad (B,A,6)])"""
    yt, pt = _f(y_true)
    yp, pp = _f(y_pred)
    B, A = yp.shape[0], yp.shape[1]
    assert yt.shape == yp.shape == (B, A, 6)
    total = np.empty((B,), np.float32)
    keep = np.empty((B, A), np.uint8)
    cls = np.empty((B, A), np.float32)
    g = np.empty((B, A, 6), np.float32) if grad else None
    npos = ctypes.c_double(0.0)
    rc = lib().orc_ssd_loss(pt, pp, B, A, int(neg_pos_ratio), int(n_neg_min), float(alpha),
                            total.ctypes.data_as(c_f), keep.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)),
                            cls.ctypes.data_as(c_f), g.ctypes.data_as(c_f) if grad else None,
                            ctypes.byref(npos))
    if rc != 0:
        raise MemoryError("orc_ssd_loss")
    res = (total, keep.astype(bool), cls, npos.value)
    return res + (g,) if grad else res
