"""Batched entry points of the SSD box pipeline (one set of launches per batch).

match_encode_batch replaces a Python loop of compute_target (reference
utils/box_utils.py:47-98, called per image at utils/datasets.py:152); decode_nms_batch
replaces the per-image post-processing block of reference inference.py:33-83.  Results
equal that loop image by image.

Arrays may be numpy arrays, tf.Tensors, anything exporting __dlpack__, or torch tensors; PyTorch
is touched only in the last case (runtime.Session).
"""
import functools

import numpy as np

from . import _lib
from .runtime import AnchorGrid, Session, _is_tf_tensor, _is_torch_tensor, _ptr, _torch


def _on_context_stream(fn):
    """With context=None a call runs on the current stream (torch tensors) or on a stream of the
    library's own (everything else).  A context given by the caller may be bound to another torch
    stream: the kernels run there, so everything else the call does must too.  The context's
    stream first waits for the current one (inputs), the whole body -- allocations, copies,
    launches, copies back to the host -- runs with the context's stream current, and the current
    stream then waits for it, with the returned CUDA tensors marked as used there."""
    @functools.wraps(fn)
    def wrapper(*args, context=None, **kwargs):
        if context is None or isinstance(context.stream, str) or not context.is_torch_stream:
            # (a raw hipStream_t value: the caller orders that stream against its own work; the
            # call keeps its arrays in library memory on it and never needs torch)
            return fn(*args, context=context, **kwargs)
        torch = _torch()
        current = torch.cuda.current_stream(context.device)
        stream = context.stream
        handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        if handle == current.cuda_stream:
            return fn(*args, context=context, **kwargs)
        stream.wait_stream(current)
        with torch.cuda.stream(stream):
            res = fn(*args, context=context, **kwargs)
        current.wait_stream(stream)
        for r in (res if isinstance(res, tuple) else (res,)):
            if isinstance(r, torch.Tensor) and r.is_cuda:
                r.record_stream(current)
        return res
    return wrapper


def _anchor_source(sess, anchors):
    """(geometry handle or None, anchor array or None, A, keepalive)"""
    if isinstance(anchors, AnchorGrid):
        if anchors.device != sess.device:
            raise ValueError("anchors were generated for GPU %d, this call runs on GPU %d"
                             % (anchors.device, sess.device))
        return anchors.geometry.handle, None, anchors.shape[0], anchors
    t, _ = sess.put(anchors, np.float32)
    if t.dim() != 2 or t.shape[1] != 4:
        raise ValueError("default boxes must have shape (A, 4), got %s" % (tuple(t.shape),))
    return None, t, int(t.shape[0]), t


@_on_context_stream
def match_encode_batch(anchors, gt_boxes, gt_count, gt_labels=None, max_size=512.0,
                       iou_threshold=0.5, y_true=False, return_match=False, context=None,
                       _device_results=False):
    """Training targets for a batch.

    anchors: AnchorGrid (on the fly) or (A,4) [cx,cy,w,h]; gt_boxes (B,G,4) normalised
    corners, zero padded; gt_count (B,) boxes per image, each in [1, G]; gt_labels (B,G)
    (default: all ones, as reference utils/datasets.py:130).
    Returns (conf (B,A), locs (B,A,4)), or y_true (B,A,6) = [one_hot(conf), locs] when
    y_true=True (reference utils/datasets.py:155-159); with return_match also
    (match_idx (B,A) int32, match_iou (B,A)) after the forced best-anchor scatter.
    Outputs are torch CUDA tensors when the inputs were torch tensors, numpy otherwise.
    """
    sess = Session(anchors, gt_boxes, gt_labels, context=context)
    ctx = sess.ctx
    if _is_torch_tensor(gt_count):
        gt_count = gt_count.cpu()
    elif _is_tf_tensor(gt_count):
        gt_count = gt_count.numpy()
    counts_host = np.asarray(gt_count)
    boxes, family = sess.put(gt_boxes, np.float32)
    if boxes.dim() != 3 or boxes.shape[2] != 4:
        raise ValueError("gt_boxes must have shape (B, G, 4), got %s" % (tuple(boxes.shape),))
    B, G = int(boxes.shape[0]), int(boxes.shape[1])
    if counts_host.shape != (B,):
        raise ValueError("gt_count must have shape (%d,), got %s" % (B, counts_host.shape))
    if B == 0 or G == 0 or counts_host.min() < 1 or counts_host.max() > G:
        # the reference drops images without boxes before this call (utils/datasets.py:84-96)
        raise ValueError("every image needs between 1 and %d ground-truth boxes" % G)
    counts = sess.from_host(counts_host.astype(np.int32))
    if gt_labels is None:
        labels = sess.from_host(np.ones((B, G), np.float32))
    else:
        labels, _ = sess.put(gt_labels, np.float32)
        if tuple(labels.shape) != (B, G):
            raise ValueError("gt_labels must have shape (%d, %d)" % (B, G))
    geom, anc, A, _keep = _anchor_source(sess, anchors)
    lib = ctx.lib
    if y_true:
        out = sess.empty((B, A, 6))
        _lib.check(lib.ssdbox_match_encode_ytrue(
            ctx.handle, geom, _ptr(anc), A, _ptr(boxes), _ptr(labels), _ptr(counts), B, G,
            float(max_size), float(iou_threshold), _ptr(out)))
        return sess.get(out, family)
    conf = sess.empty((B, A))
    locs = sess.empty((B, A, 4))
    midx = sess.empty((B, A), np.int32) if return_match else None
    miou = sess.empty((B, A)) if return_match else None
    _lib.check(lib.ssdbox_match_encode(
        ctx.handle, geom, _ptr(anc), A, _ptr(boxes), _ptr(labels), _ptr(counts), B, G,
        float(max_size), float(iou_threshold), _ptr(conf), _ptr(locs), _ptr(midx), _ptr(miou)))
    if _device_results:                 # for callers inside the package that convert themselves
        return conf, locs
    res = (sess.get(conf, family), sess.get(locs, family))
    if return_match:
        res += (sess.get(midx, family), sess.get(miou, family))
    return res


@_on_context_stream
def decode_nms_batch(anchors, y_pred, score_threshold, nms_threshold=0.45, limit=1000,
                     img_wh=None, context=None, from_logits=False):
    """Detections for a batch of head outputs.

    y_pred (B,A,6) = [softmax_bg, softmax_face, tx, ty, tw, th], or with from_logits=True the
    two class columns before the softmax of reference models/ssd.py:95 (the scores are then
    the float32 two-class softmax of those columns); img_wh (B,2) original
    (width, height) per image or None for normalised output.  Returns
    (dets (B,limit,6) rows [1, score, xmin, ymin, xmax, ymax] in kept order and zero
    after the count, keep_idx (B,limit) int32 anchor indices (-1 after the count),
    count (B,) int32).
    """
    sess = Session(anchors, y_pred, img_wh, context=context)
    ctx = sess.ctx
    y, family = sess.put(y_pred, np.float32)
    if y.dim() != 3 or y.shape[2] != 6:
        raise ValueError("y_pred must have shape (B, A, 6), got %s" % (tuple(y.shape),))
    B = int(y.shape[0])
    geom, anc, A, _keep = _anchor_source(sess, anchors)
    if int(y.shape[1]) != A:
        raise ValueError("y_pred has %d boxes per image, the anchors %d" % (int(y.shape[1]), A))
    limit = int(limit)
    wh = None
    if img_wh is not None:
        wh, _ = sess.put(img_wh, np.float32)
        if tuple(wh.shape) != (B, 2):
            raise ValueError("img_wh must have shape (%d, 2)" % B)
    dets = sess.empty((B, max(limit, 1), 6))
    keep = sess.empty((B, max(limit, 1)), np.int32)
    count = sess.empty((B,), np.int32)
    if B == 0:
        return sess.get(dets, family), sess.get(keep, family), sess.get(count, family)
    fn = ctx.lib.ssdbox_decode_nms_logits if from_logits else ctx.lib.ssdbox_decode_nms
    _lib.check(fn(
        ctx.handle, geom, _ptr(anc), A, _ptr(y), B, float(score_threshold), float(nms_threshold),
        limit, _ptr(wh), _ptr(dets), _ptr(keep), _ptr(count)))
    return sess.get(dets, family), sess.get(keep, family), sess.get(count, family)


@_on_context_stream
def prepare_gt_batch(raw_xywh, gt_count, img_wh, target_hw, shrink=None, context=None):
    """Annotation boxes of a batch to the normalised corners match_encode_batch takes.

    raw_xywh (B,G,4) WiderFace (x, y, w, h) in pixels, zero padded; gt_count (B,); img_wh (B,2)
    int original (width, height); target_hw (height, width) of the network input; shrink None
    or (B,4) int rows [h2, w2, h_off, w_off] of random_shrink (h2 <= 0: that image was not
    shrunk).  Equals, image by image, reference utils/datasets.py:131-140 followed by the box
    halves of utils/transforms.py:60-63 and :28-33.  Returns (B,G,4) float32, zero after the
    count.
    """
    sess = Session(raw_xywh, gt_count, img_wh, shrink, context=context)
    ctx = sess.ctx
    raw, family = sess.put(raw_xywh, np.float32)
    if raw.dim() != 3 or raw.shape[2] != 4:
        raise ValueError("raw_xywh must have shape (B, G, 4), got %s" % (tuple(raw.shape),))
    B, G = int(raw.shape[0]), int(raw.shape[1])
    counts, _ = sess.put(gt_count, np.int32)
    wh, _ = sess.put(img_wh, np.int32)
    if tuple(counts.shape) != (B,) or tuple(wh.shape) != (B, 2):
        raise ValueError("gt_count must have shape (%d,) and img_wh (%d, 2)" % (B, B))
    if B and int(sess.host(wh).min()) < 1:
        raise ValueError("image sizes must be positive")
    sh = None
    if shrink is not None:
        sh, _ = sess.put(shrink, np.int32)
        if tuple(sh.shape) != (B, 4):
            raise ValueError("shrink must have shape (%d, 4)" % B)
    out = sess.empty((B, G, 4))
    if B and G:
        _lib.check(ctx.lib.ssdbox_prepare_gt(ctx.handle, _ptr(raw), _ptr(counts), _ptr(wh), _ptr(sh), B, G,
                                             int(target_hw[0]), int(target_hw[1]), _ptr(out)))
    return sess.get(out, family)


@_on_context_stream
def widerface_rows_batch(dets, count, context=None):
    """Detections of decode_nms_batch (pixels of the original image) to the integer rows of
    WiderFace result files, reference evaluation.py:69-76.  dets (B,limit,6), count (B,) ->
    (xywh (B,limit,4) int32, scores (B,limit)), zero after the count."""
    sess = Session(dets, count, context=context)
    ctx = sess.ctx
    d, family = sess.put(dets, np.float32)
    if d.dim() != 3 or d.shape[2] != 6:
        raise ValueError("dets must have shape (B, limit, 6), got %s" % (tuple(d.shape),))
    B, limit = int(d.shape[0]), int(d.shape[1])
    c, _ = sess.put(count, np.int32)
    if tuple(c.shape) != (B,):
        raise ValueError("count must have shape (%d,)" % B)
    xywh = sess.empty((B, limit, 4), np.int32)
    scores = sess.empty((B, limit))
    if B and limit:
        _lib.check(ctx.lib.ssdbox_widerface_rows(ctx.handle, _ptr(d), _ptr(c), B, limit, _ptr(xywh), _ptr(scores)))
    return sess.get(xywh, family), sess.get(scores, family)


@_on_context_stream
def ssd_loss_batch(y_true, y_pred, neg_pos_ratio=3, n_neg_min=0, alpha=1.0, return_keep=False,
                   return_grad=False, context=None):
    """SSD loss with hard-negative mining, reference utils/losses.py:97-181 (SSDLoss.call).

    y_true, y_pred (B,A,6) = [background, face, 4 offsets].  Returns the (B,) per-image loss;
    with return_keep also the (B,A) bool mask of mined negatives; with return_grad also the
    (B,A,6) gradient of mean(loss) with respect to y_pred (what Keras back-propagates).
    """
    sess = Session(y_true, y_pred, context=context)
    ctx = sess.ctx
    yt, family = sess.put(y_true, np.float32)
    yp, _ = sess.put(y_pred, np.float32)
    if yt.dim() != 3 or yt.shape[2] != 6 or tuple(yt.shape) != tuple(yp.shape):
        raise ValueError("y_true and y_pred must both have shape (B, A, 6), got %s and %s"
                         % (tuple(yt.shape), tuple(yp.shape)))
    B, A = int(yt.shape[0]), int(yt.shape[1])
    if B == 0 or A == 0:
        raise ValueError("empty batch")
    total = sess.empty((B,))
    keep = sess.empty((B, A), np.uint8) if return_keep else None
    grad = sess.empty((B, A, 6)) if return_grad else None
    _lib.check(ctx.lib.ssdbox_ssd_loss(ctx.handle, _ptr(yt), _ptr(yp), B, A, int(neg_pos_ratio), int(n_neg_min),
                                       float(alpha), _ptr(total), _ptr(keep), _ptr(grad)))
    res = (sess.get(total, family),)
    if return_keep:
        k = sess.get(keep, family)
        res += (k.bool() if family == "torch" else (k != 0),)
    if return_grad:
        res += (sess.get(grad, family),)
    return res[0] if len(res) == 1 else res
