from face_detection_amd.utils.anchor_utils import generate_anchors  # noqa: F401
