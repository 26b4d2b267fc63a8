import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# SSD_CONFIG of the reference (config.py:52-69) at a chosen image size, plus the model's
# default ratio set (models/ssd.py:24-27) used when no config overrides it.
def ssd_config(size, ratios="config"):
    if ratios == "config":
        per_layer = [[0.9, 1.8, 2.5]] * 4
    else:
        per_layer = [[1.0, 2.0, 0.5]] * 4
    return {
        'img_height': size, 'img_width': size, 'image_size': (size, size),
        'offsets': 0.5, 'n_classes': 1,
        'scales': [0.007, 0.03, 0.08, 0.28, 0.60],
        'aspect_ratios_per_layer': [list(r) for r in per_layer],
        'two_boxes_for_ar1': True,
        'steps': [4, 8, 16, 32],
        'variances': [0.1, 0.2],
    }
