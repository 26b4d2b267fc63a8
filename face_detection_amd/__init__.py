"""MI355X-native SSD box pipeline: anchors, IoU matching and offset encoding for training
targets; decode, score threshold, top-k and NMS for inference.  HIP kernels behind a C ABI
(include/ssdbox.h); `face_detection_amd.utils.{anchor_utils,box_utils}` keep the reference's
function names and signatures, `match_encode_batch` / `decode_nms_batch` are the batched
forms.  Importing the package does not need a GPU; calling into it does."""
from .batched import (decode_nms_batch, match_encode_batch, prepare_gt_batch,  # noqa: F401
                      ssd_loss_batch, widerface_rows_batch)
from .runtime import AnchorGrid, Context, default_context          # noqa: F401

__all__ = ["match_encode_batch", "decode_nms_batch", "prepare_gt_batch", "widerface_rows_batch",
           "ssd_loss_batch",
           "AnchorGrid", "Context", "default_context"]
