"""Seeded synthetic WiderFace-shaped inputs for the SSD box pipeline (bench and tests).

There is no dataset or checkpoint here, so both halves of the path run on generated data
with the shapes the reference feeds them: padded ground-truth boxes per image for the
training side (reference utils/datasets.py:127-152 produces normalised corner boxes and
all-ones labels) and `(B, A, 6)` head outputs `[softmax_bg, softmax_face, tx, ty, tw, th]`
for the inference side (reference models/ssd.py:95-99).  numpy only.
"""
import math

import numpy as np

F32 = np.float32


def reference_ssd_config(size, ratios="config"):
    """SSD_CONFIG of the reference (config.py:52-69) at image side `size`.

    ratios="config" -> [0.9, 1.8, 2.5] per layer (3 boxes per cell);
    ratios="model"  -> [1.0, 2.0, 0.5] (models/ssd.py:24-27; 4 boxes per cell with the
    second aspect-ratio-1 box).
    """
    per_layer = [0.9, 1.8, 2.5] if ratios == "config" else [1.0, 2.0, 0.5]
    return {
        'img_height': size, 'img_width': size, 'image_size': (size, size),
        'offsets': 0.5, 'n_classes': 1, 'min_scale': None, 'max_scale': None,
        'scales': [0.007, 0.03, 0.08, 0.28, 0.60],
        'aspect_ratios_per_layer': [list(per_layer) for _ in range(4)],
        'two_boxes_for_ar1': True,
        'steps': [4, 8, 16, 32],
        'variances': [0.1, 0.2],
    }


def layer_table(config):
    """Per layer: (fm_w, fm_h, [(w, h), ...], first anchor index).  float64 shapes."""
    out = []
    base = 0
    for m, step in enumerate(config['steps']):
        fm_w = config['img_width'] // step
        fm_h = config['img_height'] // step
        s = config['scales']
        shapes = []
        for r in config['aspect_ratios_per_layer'][m]:
            if r == 1.0:
                shapes.append((s[m], s[m]))
                if config['two_boxes_for_ar1']:
                    q = math.sqrt(s[m] * s[m + 1])
                    shapes.append((q, q))
            else:
                shapes.append((s[m] * math.sqrt(r), s[m] / math.sqrt(r)))
        out.append((fm_w, fm_h, shapes, base))
        base += fm_w * fm_h * len(shapes)
    return out, base


def num_anchors(config):
    return layer_table(config)[1]


def make_ground_truth(rng, batch, size, gmax=256, stress=False):
    """Padded GT for `batch` images: boxes (B,gmax,4) f32 corners in [0,1], labels
    (B,gmax) f32 (1 for real boxes, 0 padding), counts (B,) int32.

    Count ~ clip(round(lognormal(ln 6, 1.2)), 1, gmax) (or == gmax when stress); side
    log-uniform in [6 px, 0.6*size], aspect w/h ~ U(0.65, 1), centre uniform with the box
    inside the image.
    """
    boxes = np.zeros((batch, gmax, 4), F32)
    labels = np.zeros((batch, gmax), F32)
    counts = np.zeros((batch,), np.int32)
    for b in range(batch):
        if stress:
            g = gmax
        else:
            g = int(np.clip(round(rng.lognormal(math.log(6.0), 1.2)), 1, gmax))
        side = np.exp(rng.uniform(math.log(6.0), math.log(0.6 * size), g))
        aspect = rng.uniform(0.65, 1.0, g)
        h = side
        w = side * aspect
        x0 = rng.uniform(0, 1, g) * (size - w)
        y0 = rng.uniform(0, 1, g) * (size - h)
        bx = np.stack([x0, y0, x0 + w, y0 + h], 1) / size
        boxes[b, :g] = bx.astype(F32)
        labels[b, :g] = 1.0
        counts[b] = g
    return boxes, labels, counts


def _softmax2(logit_face):
    # two-class softmax with the background logit fixed at 0
    e = np.exp(-np.abs(logit_face))
    hi = 1.0 / (1.0 + e)
    lo = e / (1.0 + e)
    face = np.where(logit_face >= 0, hi, lo)
    return np.stack([1.0 - face, face], -1)


def make_head_outputs(rng, config, gt_boxes, gt_counts, n_ties=64, dup_per_gt=5):
    """y_pred (B,A,6) f32: offsets ~ N(0,1); face-minus-background logit ~ N(-4, 0.55)
    through a 2-class softmax, which puts 2-3 % of the anchors above the reference's 0.05
    confidence threshold (config.py:47); around every GT `dup_per_gt` anchors get a high
    score and offsets that decode close to the GT (so NMS has clusters to suppress);
    `n_ties` random anchors per image get the exact score 1.0f.
    """
    layers, A = layer_table(config)
    B = gt_boxes.shape[0]
    y = np.empty((B, A, 6), F32)
    y[:, :, 2:] = rng.standard_normal((B, A, 4), dtype=np.float32)
    y[:, :, :2] = _softmax2(rng.normal(-4.0, 0.55, (B, A))).astype(F32)
    scale_of_layer = np.array([math.sqrt(sh[0][0] * sh[0][1]) for (_, _, sh, _) in layers])
    for b in range(B):
        for g in range(int(gt_counts[b])):
            x0, y0, x1, y1 = (float(v) for v in gt_boxes[b, g])
            gw, gh = x1 - x0, y1 - y0
            m = int(np.argmin(np.abs(np.log(scale_of_layer) - math.log(math.sqrt(gw * gh)))))
            fm_w, fm_h, shapes, base = layers[m]
            nb = len(shapes)
            cj = min(max(int((x0 + x1) / 2 * fm_w), 0), fm_w - 1)
            ci = min(max(int((y0 + y1) / 2 * fm_h), 0), fm_h - 1)
            for d in range(dup_per_gt):
                j = min(max(cj + (d % 3) - 1, 0), fm_w - 1)
                i = min(max(ci + (d // 3), 0), fm_h - 1)
                k = d % nb
                a = base + (i * fm_w + j) * nb + k
                acx, acy = (j + config['offsets']) / fm_w, (i + config['offsets']) / fm_h
                aw, ah = min(shapes[k][0], 1.0), min(shapes[k][1], 1.0)
                jit = rng.normal(0, 0.03, 4)
                bx0, by0 = x0 + jit[0] * gw, y0 + jit[1] * gh
                bx1, by1 = x1 + jit[2] * gw, y1 + jit[3] * gh
                bw, bh = max(bx1 - bx0, 1e-4), max(by1 - by0, 1e-4)
                y[b, a, 2] = ((bx0 + bx1) / 2 - acx) / (aw * 0.1)
                y[b, a, 3] = ((by0 + by1) / 2 - acy) / (ah * 0.1)
                y[b, a, 4] = math.log(bw / aw) / 0.2
                y[b, a, 5] = math.log(bh / ah) / 0.2
                p = F32(rng.uniform(0.6, 0.999))
                y[b, a, 0], y[b, a, 1] = F32(1.0) - p, p
        ties = rng.choice(A, size=min(n_ties, A), replace=False)
        y[b, ties, 0] = 0.0
        y[b, ties, 1] = 1.0
    return y
