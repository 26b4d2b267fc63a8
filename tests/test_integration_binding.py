"""A stand-alone ctypes binding of the kind a maintainer of the reference would write (no
face_detection_amd import: only the shared library and the header's signatures), run against
the numpy oracle."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ssd_config
from oracle import box_oracle as O
from oracle import c_oracle as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_standalone_ctypes_binding_of_decode_nms():
    import torch
    lib = ctypes.CDLL(os.path.join(ROOT, "face_detection_amd", "libssdbox.so"))
    vp, i32p, f64p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)
    lib.ssdbox_ctx_create.argtypes = [ctypes.c_int, vp, ctypes.c_int, ctypes.POINTER(vp)]
    lib.ssdbox_geom_create.argtypes = [ctypes.c_int] * 4 + [i32p, f64p, i32p, f64p, ctypes.c_int,
                                                            ctypes.c_double, ctypes.POINTER(vp)]
    lib.ssdbox_decode_nms.argtypes = [vp, vp, vp, ctypes.c_int64, vp, ctypes.c_int, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_int, vp, vp, vp, vp]
    lib.ssdbox_geom_num_anchors.argtypes = [vp]
    lib.ssdbox_geom_num_anchors.restype = ctypes.c_int64
    lib.ssdbox_ctx_synchronize.argtypes = [vp]
    lib.ssdbox_ctx_destroy.argtypes = [vp]
    lib.ssdbox_geom_destroy.argtypes = [vp]
    lib.ssdbox_last_error.restype = ctypes.c_char_p

    ctx, geom = vp(), vp()
    assert lib.ssdbox_ctx_create(0, None, 1, ctypes.byref(ctx)) == 0, lib.ssdbox_last_error()
    cfg = ssd_config(96)
    n = len(cfg['steps'])
    ratios = cfg['aspect_ratios_per_layer']
    assert lib.ssdbox_geom_create(
        0, cfg['img_width'], cfg['img_height'], n,
        (ctypes.c_int32 * n)(*cfg['steps']), (ctypes.c_double * (n + 1))(*cfg['scales']),
        (ctypes.c_int32 * n)(*[len(r) for r in ratios]),
        (ctypes.c_double * sum(map(len, ratios)))(*[x for r in ratios for x in r]),
        int(cfg['two_boxes_for_ar1']), cfg['offsets'], ctypes.byref(geom)) == 0, lib.ssdbox_last_error()
    anchors = O.generate_anchors(cfg)
    A, B = len(anchors), 2
    assert lib.ssdbox_geom_num_anchors(geom) == A

    rng = np.random.default_rng(4)
    y = np.empty((B, A, 6), np.float32)
    y[..., 1] = rng.beta(0.4, 3.0, (B, A))
    y[..., 0] = 1 - y[..., 1]
    y[..., 2:] = rng.normal(0, 1, (B, A, 4))
    d_y = torch.from_numpy(y).cuda()
    d_dets = torch.empty((B, 1000, 6), dtype=torch.float32, device="cuda")
    d_keep = torch.empty((B, 1000), dtype=torch.int32, device="cuda")
    d_count = torch.empty((B,), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    rc = lib.ssdbox_decode_nms(ctx, geom, None, A, d_y.data_ptr(), B, 0.05, 0.45, 1000, None,
                               d_dets.data_ptr(), d_keep.data_ptr(), d_count.data_ptr())
    assert rc == 0, lib.ssdbox_last_error()
    assert lib.ssdbox_ctx_synchronize(ctx) == 0
    for b in range(B):
        c_dets, c_idx = C.postprocess(y[b], anchors, 0.05, 1.0, 1.0, 0.45, 1000)
        k = int(d_count[b])
        assert k == len(c_idx) and k > 0
        np.testing.assert_array_equal(d_keep[b, :k].cpu().numpy(), c_idx)
        np.testing.assert_array_equal(d_dets[b, :k].cpu().numpy().view(np.uint32), c_dets.view(np.uint32))
    # an invalid call reports through the status code and the error string, nothing is thrown
    assert lib.ssdbox_decode_nms(ctx, geom, None, A, d_y.data_ptr(), B, 0.05, 1.5, 1000, None,
                                 d_dets.data_ptr(), d_keep.data_ptr(), d_count.data_ptr()) != 0
    assert b"nms_threshold" in lib.ssdbox_last_error()
    assert lib.ssdbox_ctx_destroy(ctx) == 0
    lib.ssdbox_geom_destroy(geom)
