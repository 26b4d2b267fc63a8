"""CPU-only checks: the C-ABI library loads without a GPU and exports every symbol that
include/ssdbox.h declares; host-side geometry and sharding logic."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, ssd_config
from oracle import box_oracle as O
from face_detection_amd import _lib, dist as fdist, synthetic as S
from face_detection_amd.runtime import Geometry


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ssdbox.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssdbox_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = header_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), "libssdbox.so does not export %s" % name
    assert sorted(_lib.SIGNATURES) == names       # the Python binding covers the header exactly


def test_version_and_error_string_without_gpu():
    lib = _lib.load()
    assert lib.ssdbox_version() == 100
    out = ctypes.c_void_p()
    rc = lib.ssdbox_geom_create(-1, 64, 64, 99, None, None, None, None, 1, 0.5, ctypes.byref(out))
    assert rc == _lib.ERR_INVALID and "NULL" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


@pytest.mark.parametrize("size,ratios", [(512, "config"), (640, "config"), (640, "model"),
                                         (704, "config"), (1024, "config"), (1024, "model")])
def test_host_geometry_counts(size, ratios):
    cfg = ssd_config(size, ratios)
    g = Geometry(cfg, -1)
    nb = 3 if ratios == "config" else 4
    assert g.layers() == [(size // s, size // s, nb) for s in (4, 8, 16, 32)]
    assert g.num_anchors == sum((size // s) ** 2 for s in (4, 8, 16, 32)) * nb
    assert g.num_anchors == S.num_anchors(cfg)


def test_host_geometry_matches_oracle_count_on_odd_config():
    cfg = {'img_height': 100, 'img_width': 75, 'offsets': 0.3, 'two_boxes_for_ar1': False,
           'scales': [0.1, 0.35, 0.9, 1.4], 'steps': [8, 16, 30],
           'aspect_ratios_per_layer': [[1.0, 3.0], [0.5], [1.0, 2.0, 0.5, 3.0]]}
    g = Geometry(cfg, -1)
    assert g.num_anchors == len(O.generate_anchors(cfg))
    assert g.layers() == [(9, 12, 2), (4, 6, 1), (2, 3, 4)]


def test_bad_configs_raise():
    cfg = ssd_config(64)
    cfg['scales'] = cfg['scales'][:-1]
    with pytest.raises(ValueError):
        Geometry(cfg, -1)
    cfg = ssd_config(64)
    cfg['steps'] = [4, 8, 0, 32]
    with pytest.raises(ValueError):
        Geometry(cfg, -1)
    cfg = ssd_config(64)
    cfg['aspect_ratios_per_layer'] = [[0.5 + i * 0.01 for i in range(17)]] * 4
    with pytest.raises(ValueError):
        Geometry(cfg, -1)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from face_detection_amd.utils import anchor_utils, box_utils
    with pytest.raises(RuntimeError):
        anchor_utils.generate_anchors(ssd_config(64))
    with pytest.raises(RuntimeError):
        box_utils.compute_nms(np.zeros((1, 4), np.float32), np.zeros((1,), np.float32), 0.45)


def test_shard_range_partitions():
    for n in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 4, 8):
            spans = [fdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        fdist.shard_range(8, 2, 2)


def test_synthetic_generator_shapes_and_statistics():
    rng = np.random.default_rng(0)
    cfg = ssd_config(128)
    boxes, labels, counts = S.make_ground_truth(rng, 64, 128, gmax=256)
    assert boxes.dtype == np.float32 and boxes.shape == (64, 256, 4)
    assert counts.min() >= 1 and counts.max() <= 256
    for b in range(64):
        g = counts[b]
        assert (boxes[b, :g, 2:] > boxes[b, :g, :2]).all() and (boxes[b, g:] == 0).all()
        assert boxes[b, :g].min() >= 0 and boxes[b, :g].max() <= 1
        assert labels[b, :g].all() and not labels[b, g:].any()
    cfg = ssd_config(640)
    y = S.make_head_outputs(rng, cfg, boxes[:2], counts[:2])
    assert y.shape == (2, 102000, 6) and y.dtype == np.float32
    assert np.allclose(y[..., 0] + y[..., 1], 1, atol=1e-6)
    frac = (y[..., 1] > 0.05).mean()
    assert 0.015 < frac < 0.04
    assert ((y[..., 1] == 1.0).sum(1) >= 64).all()


def test_utils_shim_leaves_the_other_reference_modules_importable(tmp_path):
    """`utils` at the repo root shadows only anchor_utils / box_utils: a stand-in for the
    reference tree (own files, not the reference's) keeps its utils.datasets importable, and
    that module's `from utils.box_utils import ...` lands here."""
    import subprocess
    import sys
    ref = tmp_path / "ref" / "utils"
    ref.mkdir(parents=True)
    (ref / "datasets.py").write_text("from utils.box_utils import compute_target\nMARK = 'ref-datasets'\n")
    (ref / "box_utils.py").write_text("MARK = 'ref-box'\n")
    code = ("import utils.datasets as d, utils.box_utils as b, utils.anchor_utils as a, os;"
            "assert d.MARK == 'ref-datasets', d.__file__;"
            "assert not hasattr(b, 'MARK') and callable(b.compute_target) and callable(a.generate_anchors);"
            "assert d.compute_target is b.compute_target;"
            "assert os.path.dirname(os.path.abspath(b.__file__)) == os.path.join(%r, 'utils');"
            "print('ok')" % ROOT)
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path / "ref"), env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]
