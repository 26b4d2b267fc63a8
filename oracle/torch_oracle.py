"""Second CPU restatement of the SSD box pipeline, in PyTorch float32 -- TEST INFRASTRUCTURE, NOT
PRODUCT CODE.  Only tests/ may import it.

Purpose: an independently written cross-check of oracle/box_oracle.py (numpy) and
oracle/ssdbox_oracle.c.  It follows the same reference functions (utils/anchor_utils.py:6-62,
utils/box_utils.py:4-222, inference.py:33-83) but is organised differently on purpose -- anchors
from broadcast index grids instead of a Python loop, the training target from per-row reductions
and an explicit ordered scatter, NMS as a sweep over a precomputed suppression matrix instead of
the column-masking loop -- so that an error of reading shared by two restatements is less likely
than one made twice.  The IEEE operations (+ - * / min max compare) must agree with the numpy
oracle bit for bit; exp / log agree to a few ULP (different libm).

PARITY UNPINNED, like the other restatements: nothing here was compared with outputs of the
reference (TensorFlow is not installed where this was written).  The conventions frozen in
oracle/box_oracle.py's header hold here too (ties to the lowest index, ordered scatter with the
last duplicate winning, stable descending sort with NaN scores last, a NaN IoU never wins a
maximum)."""
import math

import torch

F32 = torch.float32


def _t(x, dtype=F32):
    return torch.as_tensor(x, dtype=dtype)


# utils/anchor_utils.py:6-62
def generate_anchors(config):
    layers = []
    scales = config['scales']
    off = float(config['offsets'])
    for m, step in enumerate(config['steps']):
        fm_w, fm_h = config['img_width'] // step, config['img_height'] // step
        wh = []
        for r in config['aspect_ratios_per_layer'][m]:                 # :35-57
            if r == 1.0:
                wh.append((scales[m], scales[m]))
                if config['two_boxes_for_ar1']:
                    s = math.sqrt(scales[m] * scales[m + 1])
                    wh.append((s, s))
            else:
                wh.append((scales[m] * math.sqrt(r), scales[m] / math.sqrt(r)))
        # centres in float64 (Python floats in the reference), one rounding to float32 at :59
        cx = (torch.arange(fm_w, dtype=torch.float64) + off) / fm_w
        cy = (torch.arange(fm_h, dtype=torch.float64) + off) / fm_h
        shapes = torch.tensor(wh, dtype=torch.float64)                 # (n_box, 2)
        n_box = shapes.shape[0]
        grid = torch.empty((fm_h, fm_w, n_box, 4), dtype=torch.float64)
        grid[..., 0] = cx.view(1, fm_w, 1)
        grid[..., 1] = cy.view(fm_h, 1, 1)
        grid[..., 2] = shapes[:, 0].view(1, 1, n_box)
        grid[..., 3] = shapes[:, 1].view(1, 1, n_box)
        layers.append(grid.reshape(-1, 4))
    return torch.cat(layers).to(F32).clamp(0.0, 1.0)                   # :59-60


# utils/box_utils.py:4-18
def compute_area(top_left, bot_right, max_size=512.0):
    side = torch.clamp(_t(bot_right) - _t(top_left), 0.0, float(max_size))
    return side.select(-1, 0) * side.select(-1, 1)


# utils/box_utils.py:21-44
def compute_iou(boxes_a, boxes_b, max_size=512.0):
    a, b = _t(boxes_a).unsqueeze(1), _t(boxes_b).unsqueeze(0)
    inter = compute_area(torch.maximum(a[..., :2], b[..., :2]), torch.minimum(a[..., 2:], b[..., 2:]), max_size)
    total = compute_area(a[..., :2], a[..., 2:], max_size) + compute_area(b[..., :2], b[..., 2:], max_size)
    return inter / (total - inter)


# utils/box_utils.py:150-181
def transform_corner_to_center(boxes):
    boxes = _t(boxes)
    lo, hi = boxes[..., :2], boxes[..., 2:]
    return torch.cat([(lo + hi) / 2, hi - lo], -1)


def transform_center_to_corner(boxes):
    boxes = _t(boxes)
    c, half = boxes[..., :2], boxes[..., 2:] / 2
    return torch.cat([c - half, c + half], -1)


# utils/box_utils.py:101-123
def encode(default_boxes, boxes, variance=(0.1, 0.2)):
    d, g = _t(default_boxes), transform_corner_to_center(boxes)
    v0, v1 = _t(variance[0]), _t(variance[1])
    return torch.cat([(g[..., :2] - d[:, :2]) / (d[:, 2:] * v0), torch.log(g[..., 2:] / d[:, 2:]) / v1], -1)


# utils/box_utils.py:126-147
def decode(default_boxes, locs, variance=(0.1, 0.2)):
    d, l = _t(default_boxes), _t(locs)
    v0, v1 = _t(variance[0]), _t(variance[1])
    centre = l[..., :2] * v0 * d[:, 2:] + d[:, :2]
    size = torch.exp(l[..., 2:] * v1) * d[:, 2:]
    return transform_center_to_corner(torch.cat([centre, size], -1))


# utils/box_utils.py:47-98
def compute_target(default_boxes, gt_boxes, gt_labels, max_size=512.0, iou_threshold=0.5, details=False):
    d, g = _t(default_boxes), _t(gt_boxes)
    labels = torch.as_tensor(gt_labels)
    iou = compute_iou(transform_center_to_corner(d), g, max_size)      # (A, G)
    iou = torch.where(torch.isnan(iou), torch.zeros_like(iou), iou)    # a NaN IoU never wins
    # first maximum per row / per column (tf.argmax): the lowest index among equal values
    best_iou, _ = iou.max(dim=1)
    G = iou.shape[1]
    col = torch.arange(G).view(1, G).expand_as(iou)
    best_gt = torch.where(iou == best_iou.view(-1, 1), col, torch.full_like(col, G)).min(dim=1).values
    A = iou.shape[0]
    row = torch.arange(A).view(A, 1).expand_as(iou)
    top_per_gt, _ = iou.max(dim=0)
    best_anchor = torch.where(iou == top_per_gt.view(1, -1), row, torch.full_like(row, A)).min(dim=0).values
    for k in range(G):                                                  # :72-87, updates in order
        best_gt[best_anchor[k]] = k
        best_iou[best_anchor[k]] = 1.0
    conf = labels[best_gt]
    conf = torch.where(best_iou < _t(iou_threshold), torch.zeros_like(conf), conf)
    locs = encode(d, g[best_gt])
    if details:
        return conf, locs, best_gt, best_iou
    return conf, locs


# utils/box_utils.py:184-222
def compute_nms(boxes, scores, nms_threshold, limit=200):
    boxes, scores = _t(boxes), _t(scores)
    if boxes.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.int32)
    # DESCENDING argsort as top_k: stable, NaN after every number
    key = torch.where(torch.isnan(scores), torch.full_like(scores, -math.inf), scores)
    is_nan = torch.isnan(scores).to(torch.int8)
    order = torch.arange(len(scores))
    order = order[torch.sort(-key[order], stable=True).indices]
    order = order[torch.sort(is_nan[order], stable=True).indices][:limit]
    ranked = boxes[order]
    kills = ~(compute_iou(ranked, ranked) <= _t(nms_threshold))        # row box removes column box
    alive = torch.ones(len(order), dtype=torch.bool)
    kept = []
    i = 0                                   # the reference starts from rank 0 unconditionally (:200)
    while True:
        kept.append(i)
        alive &= ~kills[i]
        alive[i] = False
        rest = torch.nonzero(alive).flatten()
        if rest.numel() == 0:
            break
        i = int(rest[0])
    return order[torch.tensor(kept)].to(torch.int32)


# inference.py:33-83
def postprocess(y_pred_image, anchors, conf_thresh, img_width=1.0, img_height=1.0, nms_threshold=0.45, limit=1000):
    y = _t(y_pred_image)
    scores = y[:, 1]
    boxes = decode(anchors, y[:, 2:])
    cand = torch.nonzero(scores > _t(conf_thresh)).flatten()
    keep = compute_nms(boxes[cand], scores[cand], nms_threshold, limit).long()
    sel = cand[keep]
    out = boxes[sel].clamp(0.0, 1.0) * _t([img_width, img_height, img_width, img_height])
    dets = torch.cat([torch.ones((len(sel), 1)), scores[sel].view(-1, 1), out], 1)
    return dets, sel.to(torch.int32)
