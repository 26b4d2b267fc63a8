from face_detection_amd.utils.box_utils import (  # noqa: F401
    compute_area, compute_iou, compute_nms, compute_target, decode, encode,
    transform_center_to_corner, transform_corner_to_center)
