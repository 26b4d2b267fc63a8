"""CPU oracle for the SSD box pipeline -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (face_detection_amd/) never does and fails loudly without its
HIP library.

What this is: a numpy float32 restatement, op for op, of the reference's TensorFlow-eager
box math (reference utils/anchor_utils.py:6-62 and utils/box_utils.py:4-222) and of the
two caller tails that define the data formats on either side of it
(utils/datasets.py:152-159, inference.py:33-83).  The "literal" functions keep the
reference's data flow on purpose -- Python anchor loop, materialised (A,G,2) IoU
temporaries, two full reductions, scatter, encode for every anchor, NMS as an O(K^3)
column-masking loop -- because they double as the timed CPU baseline.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path,
and TensorFlow (its only runtime) is not installed where this was written, so nothing
here was checked against reference outputs.  It is pinned only by hand-derived
known-answer tests (tests/test_oracle_kat.py).  TF behaviours frozen by convention:
  * tf.constant(list of Python floats) is float32; all arithmetic is IEEE float32;
  * tf.argmax ties resolve to the lowest index;
  * tf.tensor_scatter_nd_update on CPU applies updates in order (last duplicate wins);
  * tf.argsort(direction='DESCENDING') is stable (lower index first on equal scores);
  * comparisons with NaN are false; a NaN IoU (0/0 of two zero-area boxes) never wins a
    max/argmax (it is treated as 0 there); NaN scores rank last;
  * exp/log are whatever the float32 libm-quality routine returns (numpy's here), so
    encoded offsets / decoded boxes are comparable to 1e-5 relative, not bit for bit.
"""
import math

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- anchors
def layer_box_shapes(config, m):
    """(w, h) of every box of one cell of layer m, in emission order, as Python floats.

    Follows utils/anchor_utils.py:35-57: a ratio equal to 1.0 emits the square box of
    scale s_m and, if two_boxes_for_ar1, a second square of sqrt(s_m * s_{m+1}); any
    other ratio emits (s_m * sqrt(r), s_m / sqrt(r)).
    """
    scales = config['scales']
    shapes = []
    for ratio in config['aspect_ratios_per_layer'][m]:
        if ratio == 1.0:
            shapes.append((scales[m], scales[m]))
            if config['two_boxes_for_ar1']:
                side = math.sqrt(scales[m] * scales[m + 1])
                shapes.append((side, side))
        else:
            root = math.sqrt(ratio)
            shapes.append((scales[m] * root, scales[m] / root))
    return shapes


def generate_anchors(config):
    """All default boxes, (A,4) float32 [cx, cy, w, h] clipped to [0,1].

    Follows utils/anchor_utils.py:6-62: layer-major, then row (y), column (x), box;
    centres are (index + offsets) / feature-map size in float64, rounded once to float32
    (:59) and clipped afterwards (:60).
    """
    off = config['offsets']
    rows = []
    for m, step in enumerate(config['steps']):
        fm_w = config['img_width'] // step
        fm_h = config['img_height'] // step
        shapes = layer_box_shapes(config, m)
        for i in range(fm_h):
            cy = (i + off) / fm_h
            for j in range(fm_w):
                cx = (j + off) / fm_w
                for (w, h) in shapes:
                    rows.append([cx, cy, w, h])
    anchors = np.array(rows, dtype=np.float32).reshape(-1, 4)
    return np.clip(anchors, F32(0.0), F32(1.0))


# ------------------------------------------------------------------ elementwise boxes
def compute_area(top_left, bot_right, max_size=512.0):
    """utils/box_utils.py:4-18."""
    hw = np.clip(np.asarray(bot_right, F32) - np.asarray(top_left, F32), F32(0.0), F32(max_size))
    return hw[..., 0] * hw[..., 1]


def compute_iou(boxes_a, boxes_b, max_size=512.0):
    """utils/box_utils.py:21-44 -- (Na,4),(Nb,4) corners -> (Na,Nb), no epsilon."""
    a = np.asarray(boxes_a, F32)[:, None, :]
    b = np.asarray(boxes_b, F32)[None, :, :]
    top_left = np.maximum(a[..., :2], b[..., :2])
    bot_right = np.minimum(a[..., 2:], b[..., 2:])
    overlap = compute_area(top_left, bot_right, max_size)
    area_a = compute_area(a[..., :2], a[..., 2:], max_size)
    area_b = compute_area(b[..., :2], b[..., 2:], max_size)
    with np.errstate(divide='ignore', invalid='ignore'):
        return overlap / (area_a + area_b - overlap)


def transform_corner_to_center(boxes):
    """utils/box_utils.py:150-164."""
    boxes = np.asarray(boxes, F32)
    return np.concatenate([(boxes[..., :2] + boxes[..., 2:]) / F32(2),
                           boxes[..., 2:] - boxes[..., :2]], axis=-1)


def transform_center_to_corner(boxes):
    """utils/box_utils.py:167-181."""
    boxes = np.asarray(boxes, F32)
    half = boxes[..., 2:] / F32(2)
    return np.concatenate([boxes[..., :2] - half, boxes[..., :2] + half], axis=-1)


def encode(default_boxes, boxes, variance=None):
    """utils/box_utils.py:101-123 -- divides by (a_wh * v0), log(wh / a_wh) / v1."""
    if variance is None:
        variance = [0.1, 0.2]
    d = np.asarray(default_boxes, F32)
    t = transform_corner_to_center(boxes)
    with np.errstate(divide='ignore', invalid='ignore'):
        xy = (t[..., :2] - d[:, :2]) / (d[:, 2:] * F32(variance[0]))
        wh = np.log(t[..., 2:] / d[:, 2:]) / F32(variance[1])
    return np.concatenate([xy, wh], axis=-1).astype(F32)


def decode(default_boxes, locs, variance=None):
    """utils/box_utils.py:126-147 -- ((l*v0)*a_wh)+a_xy ; exp(l*v1)*a_wh ; to corners."""
    if variance is None:
        variance = [0.1, 0.2]
    d = np.asarray(default_boxes, F32)
    l = np.asarray(locs, F32)
    with np.errstate(over='ignore', invalid='ignore'):
        xy = l[..., :2] * F32(variance[0]) * d[:, 2:] + d[:, :2]
        wh = np.exp(l[..., 2:] * F32(variance[1])) * d[:, 2:]
        return transform_center_to_corner(np.concatenate([xy, wh], axis=-1).astype(F32))


# -------------------------------------------------------------------- training target
def match(default_boxes, gt_boxes, max_size=512.0):
    """Steps (1)-(3) of compute_target (utils/box_utils.py:63-87).

    Returns (best_gt_idx int64 (A,), best_gt_iou f32 (A,), best_default_idx int64 (G,))
    after the forced best-anchor-per-GT scatter (ascending GT order, last write wins).
    """
    iou = compute_iou(transform_center_to_corner(default_boxes), gt_boxes, max_size)
    iou = np.where(np.isnan(iou), F32(0.0), iou)          # NaN never wins (header)
    best_gt_iou = iou.max(axis=1)
    best_gt_idx = iou.argmax(axis=1)
    best_default_idx = iou.argmax(axis=0)
    for g, a in enumerate(best_default_idx):
        best_gt_idx[a] = g
    best_gt_iou[best_default_idx] = F32(1.0)
    return best_gt_idx, best_gt_iou, best_default_idx


def compute_target(default_boxes, gt_boxes, gt_labels, max_size=512.0, iou_threshold=0.5):
    """utils/box_utils.py:47-98 -> (gt_confs (A,), gt_locs (A,4)); no neutral band,
    positives are the anchors whose (forced) IoU is not < threshold, every anchor is
    encoded against its matched GT."""
    default_boxes = np.asarray(default_boxes, F32)
    gt_boxes = np.asarray(gt_boxes, F32)
    gt_labels = np.asarray(gt_labels)
    best_gt_idx, best_gt_iou, _ = match(default_boxes, gt_boxes, max_size)
    confs = gt_labels[best_gt_idx]
    confs = np.where(best_gt_iou < F32(iou_threshold), np.zeros_like(confs), confs)
    locs = encode(default_boxes, gt_boxes[best_gt_idx])
    return confs, locs


def build_y_true(default_boxes, gt_boxes, gt_labels, max_size):
    """Training-side consumer layout, utils/datasets.py:152-159: one_hot(int(conf), 2)
    concatenated with the four offsets -> (A,6) float32."""
    confs, locs = compute_target(default_boxes, gt_boxes, gt_labels, max_size)
    cls = confs.astype(np.int32)
    one_hot = np.zeros((cls.shape[0], 2), F32)
    for c in (0, 1):
        one_hot[cls == c, c] = 1.0
    return np.concatenate([one_hot, locs.astype(F32)], axis=1)


# --------------------------------------------------------------------------- inference
def argsort_descending(scores):
    """tf.argsort(direction='DESCENDING') == top_k(k=N): stable, NaN last."""
    return np.argsort(-np.asarray(scores, F32), kind='stable')


def compute_nms(boxes, scores, nms_threshold, limit=200):
    """utils/box_utils.py:184-222, literally: pre-NMS top-`limit`, K x K IoU with the
    default max_size, then the loop that kills columns of every box the last selected
    one overlaps by more than the threshold (or NaN) and picks the first survivor."""
    boxes = np.asarray(boxes, F32)
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.int32)
    order = argsort_descending(scores)[:limit]
    kept_boxes = boxes[order]
    iou = compute_iou(kept_boxes, kept_boxes)
    thr = F32(nms_threshold)
    selected = [0]
    while True:
        alive = iou[selected[-1]] <= thr
        iou = np.where(~alive[None, :], F32(1.0), iou)
        if not alive.any():
            break
        selected.append(int(np.argsort(-alive.astype(np.int32), kind='stable')[0]))
    return order[np.array(selected)].astype(np.int32)


def compute_nms_fast(boxes, scores, nms_threshold, limit=200):
    """Same result as compute_nms without rewriting the K x K matrix every step (used
    for large goldens; checked against the literal loop in the tests)."""
    boxes = np.asarray(boxes, F32)
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.int32)
    order = argsort_descending(scores)[:limit]
    kept_boxes = boxes[order]
    iou = compute_iou(kept_boxes, kept_boxes)
    dead = ~(iou <= F32(nms_threshold))            # True where row box kills column box
    alive = np.ones(len(order), bool)
    selected = []
    i = 0
    while True:
        selected.append(i)
        alive &= ~dead[i]
        alive[i] = False
        nxt = np.flatnonzero(alive)
        if nxt.size == 0:
            break
        i = int(nxt[0])
    return order[np.array(selected)].astype(np.int32)


def softmax2(y_logits):
    """Class activation of the head, reference models/ssd.py:95: softmax over the two class
    columns of (..., 6) rows [logit_bg, logit_face, 4 offsets], float32, evaluated as TF does
    (exp of the shifted logits times the reciprocal of their sum)."""
    y = np.array(y_logits, F32)
    c = y[..., :2]
    e = np.exp(c - np.max(c, axis=-1, keepdims=True)).astype(F32)
    inv = (F32(1.0) / np.sum(e, axis=-1, keepdims=True, dtype=F32)).astype(F32)
    y[..., :2] = e * inv
    return y


def postprocess(y_pred_image, anchors, conf_thresh, img_width=1.0, img_height=1.0,
                nms_threshold=0.45, limit=1000, nms=compute_nms):
    """Inference-side caller tail for one image, inference.py:33-83.

    y_pred_image is (A,6) [softmax_bg, softmax_face, tx, ty, tw, th].  Returns
    (dets (n,6) float32 rows [class=1, score, xmin, ymin, xmax, ymax] in the original
    image's pixels, anchor_idx (n,) int32 of the kept boxes in kept order).
    """
    y = np.asarray(y_pred_image, F32)
    scores = y[:, 1]
    boxes = decode(anchors, y[:, 2:])
    mask = scores > F32(conf_thresh)
    cand = np.flatnonzero(mask)
    keep = nms(boxes[cand], scores[cand], nms_threshold, limit)
    sel = cand[keep]
    out = np.clip(boxes[sel], F32(0.0), F32(1.0))
    out[:, 0] *= F32(img_width)
    out[:, 2] *= F32(img_width)
    out[:, 1] *= F32(img_height)
    out[:, 3] *= F32(img_height)
    dets = np.concatenate([np.ones((len(sel), 1), F32), scores[sel][:, None], out], axis=1)
    return dets.astype(F32), sel.astype(np.int32)


# --------------------------------------------------------------------------------------
# The data formats either side of the path (test infrastructure like the rest of this file).
# The reference evaluates these lines on float32 numpy arrays against Python ints and int32
# scalars; with the numpy of its time (value-based casting) every operation stays float32.
# The scalars are converted to float32 explicitly here so that the result does not depend on
# the numpy version running the oracle.
# --------------------------------------------------------------------------------------
def prepare_gt(raw_xywh, img_w, img_h, target_h, target_w, shrink=None):
    """WiderFace (x, y, w, h) pixel boxes of one image -> normalised corners as compute_target
    receives them: utils/datasets.py:131-140, then (shrink = (h2, w2, h_off, w_off)) the box
    half of random_shrink, utils/transforms.py:60-63, then the box half of reize_with_pad,
    utils/transforms.py:28-33."""
    boxes = np.array(raw_xywh, dtype=F32).reshape(-1, 4)
    boxes[:, 2] = boxes[:, 0] + boxes[:, 2]
    boxes[:, 3] = boxes[:, 1] + boxes[:, 3]
    boxes -= F32(1)
    w, h = F32(img_w), F32(img_h)
    boxes[:, 0] = boxes[:, 0] / w
    boxes[:, 2] = boxes[:, 2] / w
    boxes[:, 1] = boxes[:, 1] / h
    boxes[:, 3] = boxes[:, 3] / h
    if shrink is not None:
        h2, w2, h_off, w_off = (F32(v) for v in shrink)
        boxes[:, 0] = (boxes[:, 0] * w2 + w_off - F32(1)) / w
        boxes[:, 1] = (boxes[:, 1] * h2 + h_off - F32(1)) / h
        boxes[:, 2] = (boxes[:, 2] * w2 + w_off - F32(1)) / w
        boxes[:, 3] = (boxes[:, 3] * h2 + h_off - F32(1)) / h
    h1, w1 = int(img_h), int(img_w)
    if h1 / w1 < target_h / target_w:
        den, num = F32(target_h * w1), F32(h1 * target_w)
        boxes[:, 1] = (boxes[:, 1] - F32(0.5)) / den * num + F32(0.5)
        boxes[:, 3] = (boxes[:, 3] - F32(0.5)) / den * num + F32(0.5)
    else:
        den, num = F32(target_w * h1), F32(w1 * target_h)
        boxes[:, 0] = (boxes[:, 0] - F32(0.5)) / den * num + F32(0.5)
        boxes[:, 2] = (boxes[:, 2] - F32(0.5)) / den * num + F32(0.5)
    return boxes


def widerface_rows(preds):
    """evaluation.py:69-76: (n,6) detection rows in pixels -> (n,4) integer (x, y, w, h)."""
    preds = np.asarray(preds, dtype=np.float64).reshape(-1, 6)
    boxes = preds[:, 2:6].astype(np.int64)
    boxes += 1
    boxes[:, 2] = boxes[:, 2] - boxes[:, 0]
    boxes[:, 3] = boxes[:, 3] - boxes[:, 1]
    return boxes, preds[:, 1]


def widerface_lines(name, preds):
    """evaluation.py:70,77-78: the text of one result file."""
    boxes, scores = widerface_rows(preds)
    lines = [name + "\n", "%d\n" % len(boxes)]
    for box, score in zip(boxes, scores):
        lines.append("%d %d %d %d %.3f\n" % (box[0], box[1], box[2], box[3], score))
    return "".join(lines)


# --------------------------------------------------------------------------------------
# SSD loss with hard-negative mining, reference utils/losses.py:41-181 (forward), and the
# gradient TensorFlow's autodiff derives from it.  Float32 like the reference (floatx).
# Frozen conventions: tf.nn.top_k ranks equal values by ascending index; count_nonzero treats
# -0 as zero and NaN as nonzero; the mask and the counts carry no gradient.
# --------------------------------------------------------------------------------------
def smooth_l1(y_true, y_pred):
    """utils/losses.py:58-61"""
    diff = (y_true - y_pred).astype(F32)
    absolute = np.abs(diff)
    square = F32(0.5) * diff ** 2
    return np.where(absolute < F32(1.0), square, absolute - F32(0.5)).astype(F32).sum(axis=-1, dtype=F32)


def log_loss(y_true, y_pred, log=np.log):
    """utils/losses.py:78-81"""
    p = np.maximum(y_pred.astype(F32), F32(1e-15))
    return (-(y_true.astype(F32) * log(p).astype(F32)).sum(axis=-1, dtype=F32)).astype(F32)


def topk_indices(values, k):
    """tf.nn.top_k(values, k).indices as a set: the k largest, equal values by ascending index
    (NaN ranks last)."""
    v = np.asarray(values, F32)
    key = np.where(np.isnan(v), -np.inf, v.astype(np.float64))
    order = np.argsort(-key, kind="stable")
    return order[:k]


def ssd_loss(y_true, y_pred, neg_pos_ratio=3, n_neg_min=0, alpha=1.0, log=np.log, details=False):
    """utils/losses.py:97-181 for y_true, y_pred of shape (B, A, 2 + 4).  Returns the (B,)
    loss; with details also (keep mask (B,A) bool, classification loss (B,A), n_positive)."""
    yt = np.asarray(y_true, F32)
    yp = np.asarray(y_pred, F32)
    B, A = yp.shape[0], yp.shape[1]
    cls = log_loss(yt[:, :, :-4], yp[:, :, :-4], log)                    # :102
    loc = smooth_l1(yt[:, :, -4:], yp[:, :, -4:])                        # :103
    negatives = yt[:, :, 0]                                              # :108
    positives = yt[:, :, 1:-4].max(axis=-1)                              # :109
    n_positive = positives.sum(dtype=np.float64)                         # :112 (exact for 0/1 labels)
    pos_class = (cls * positives).sum(axis=-1, dtype=np.float64)         # :117
    neg_all = (cls * negatives).astype(F32)                              # :122
    n_neg_losses = int(np.count_nonzero(neg_all))                        # :123
    k = min(max(int(neg_pos_ratio) * int(n_positive), int(n_neg_min)), n_neg_losses)   # :135-136
    keep = np.zeros((B * A,), bool)
    if n_neg_losses != 0:                                                # :163
        keep[topk_indices(neg_all.reshape(-1), k)] = True                # :149-157
    keep = keep.reshape(B, A)
    neg_class = (cls * keep).sum(axis=-1, dtype=np.float64)              # :160
    loc_sum = (loc * positives).sum(axis=-1, dtype=np.float64)           # :170
    total = (pos_class + neg_class + float(alpha) * loc_sum) / max(1.0, float(n_positive))   # :174
    total = (total * B).astype(F32)                                      # :179
    if details:
        return total, keep, cls, n_positive
    return total


def ssd_loss_grad(y_true, y_pred, keep, n_positive, alpha=1.0):
    """d mean_b(loss_b) / d y_pred, what Keras back-propagates (the loss is averaged over the
    batch after utils/losses.py:179 multiplied by it): per element
    [-(y_true_c * w) / p_c where p_c >= 1e-15 | alpha * positives * smooth_l1'] / max(1, n_pos),
    w = positives + keep."""
    yt = np.asarray(y_true, F32)
    yp = np.asarray(y_pred, F32)
    scale = F32(1.0 / max(1.0, float(n_positive)))
    positives = yt[:, :, 1:-4].max(axis=-1)
    w = (positives + keep.astype(F32)).astype(F32)
    grad = np.zeros_like(yp)
    p = yp[:, :, :-4]
    gp = np.where(p >= F32(1e-15), -(yt[:, :, :-4] * w[:, :, None]) / np.maximum(p, F32(1e-15)), F32(0))
    grad[:, :, :-4] = gp * scale
    diff = yp[:, :, -4:] - yt[:, :, -4:]
    dl = np.where(np.abs(diff) < F32(1.0), diff, np.sign(diff)).astype(F32)
    grad[:, :, -4:] = F32(alpha) * positives[:, :, None] * dl * scale
    return grad.astype(F32)
