"""Drop-in for the reference's utils/anchor_utils.py (reference utils/anchor_utils.py:6-62)."""
from ..runtime import AnchorGrid


def generate_anchors(config):
    """Default boxes of every feature map for an SSD config dict with the reference's keys
    (scales, steps, two_boxes_for_ar1, aspect_ratios_per_layer, img_height, img_width,
    offsets; reference config.py:52-69).

    Returns an array-like of shape (A, 4), float32 [cx, cy, w, h] clipped to [0, 1], in the
    reference's order (layer, row, column, box).  The array is not stored: the returned
    AnchorGrid carries the feature-map geometry, compute_target / decode / the batched
    entry points build anchors from it inside their kernels, and np.asarray(), indexing or
    .tensor() materialise it on the GPU on first use.
    """
    return AnchorGrid(config)
