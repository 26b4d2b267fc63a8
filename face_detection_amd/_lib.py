"""ctypes binding of libssdbox.so (the C ABI declared in include/ssdbox.h).

There is no CPU fallback: if the library has not been built, or no MI355X is visible,
every entry point of this package raises.
"""
import ctypes
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSDBOX_LIB selects another build of the same library (kernel experiments); never a fallback.
LIB_PATH = os.environ.get("SSDBOX_LIB") or os.path.join(_HERE, "libssdbox.so")

OK = 0
ERR_INVALID = -1
ERR_HIP = -2
ERR_NOMEM = -3
ERR_UNSUPPORTED = -4

NUM_STAGES = 10
STAGE_NAMES = ("match", "force", "select", "topk", "nms_mask", "nms_scan", "elementwise", "shapes",
               "loss", "prune")
MAX_NMS_LIMIT = 4096
MAX_GT = 4096
MAX_LAYERS = 8
STATIC_SCRATCH_BYTES = 4 << 20

_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_f = ctypes.c_float
_d = ctypes.c_double
_pi32 = ctypes.POINTER(ctypes.c_int32)
_pf = ctypes.POINTER(ctypes.c_float)
_pd = ctypes.POINTER(ctypes.c_double)

# name -> (restype, argtypes); the single source of truth for the binding and for the
# symbol-export test.
SIGNATURES = {
    "ssdbox_version": (_i, []),
    "ssdbox_last_error": (ctypes.c_char_p, []),
    "ssdbox_ctx_create": (_i, [_i, _vp, _i, ctypes.POINTER(_vp)]),
    "ssdbox_ctx_destroy": (_i, [_vp]),
    "ssdbox_ctx_set_stream": (_i, [_vp, _vp, _i]),
    "ssdbox_ctx_synchronize": (_i, [_vp]),
    "ssdbox_ctx_set_timing": (_i, [_vp, _i]),
    "ssdbox_ctx_collect_timing": (_i, [_vp, _pf, _pi32]),
    "ssdbox_ctx_collect_timeline": (_i, [_vp, _vp, _pi32, _pf, _pf, _i, _pi32]),
    "ssdbox_device_count": (_i, []),
    "ssdbox_buf_alloc": (_i, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "ssdbox_buf_free": (_i, [_vp, _vp]),
    "ssdbox_copy_h2d": (_i, [_vp, _vp, _vp, ctypes.c_size_t]),
    "ssdbox_copy_d2h": (_i, [_vp, _vp, _vp, ctypes.c_size_t]),
    "ssdbox_copy_d2d": (_i, [_vp, _vp, _vp, ctypes.c_size_t]),
    "ssdbox_host_alloc": (_i, [ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "ssdbox_host_free": (_i, [_vp]),
    "ssdbox_ctx_get_stream": (_i, [_vp, ctypes.POINTER(_vp)]),
    "ssdbox_dlpack_import": (_i, [_vp, _vp, ctypes.c_int32, ctypes.POINTER(_vp), ctypes.POINTER(_i64), _pi32, _pi32,
                                  _pi32]),
    "ssdbox_dlpack_export": (_i, [_vp, _vp, ctypes.c_int32, ctypes.POINTER(_i64), ctypes.c_int32, ctypes.c_int32,
                                  ctypes.POINTER(_vp)]),
    "ssdbox_dlpack_delete": (_i, [_vp]),
    "ssdbox_geom_create": (_i, [_i, _i, _i, _i, _pi32, _pd, _pi32, _pd, _i, _d, ctypes.POINTER(_vp)]),
    "ssdbox_geom_destroy": (_i, [_vp]),
    "ssdbox_geom_num_anchors": (_i64, [_vp]),
    "ssdbox_geom_layers": (_i, [_vp, _pi32, _pi32, _pi32]),
    "ssdbox_generate_anchors": (_i, [_vp, _vp, _vp]),
    "ssdbox_match_encode": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "ssdbox_match_encode_static": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp, _vp, _vp]),
    "ssdbox_match_encode_ytrue": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _f, _f, _vp]),
    "ssdbox_decode_nms": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "ssdbox_decode_nms_logits": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "ssdbox_compute_nms": (_i, [_vp, _vp, _vp, _i64, _f, _i, _vp, _vp]),
    "ssdbox_compute_area": (_i, [_vp, _vp, _vp, _i64, _f, _vp]),
    "ssdbox_compute_iou": (_i, [_vp, _vp, _i64, _vp, _i64, _f, _vp]),
    "ssdbox_encode": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _f, _f, _vp]),
    "ssdbox_decode": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _f, _f, _vp]),
    "ssdbox_corner_to_center": (_i, [_vp, _vp, _i64, _vp]),
    "ssdbox_center_to_corner": (_i, [_vp, _vp, _i64, _vp]),
    "ssdbox_prepare_gt": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ssdbox_widerface_rows": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ssdbox_ssd_loss": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _f, _vp, _vp, _vp]),
    "ssdbox_selftest": (_i, [_vp, _pf, _pf, _pf, _pf, _i]),
}

_lib = None


class SsdboxError(RuntimeError):
    """A libssdbox call failed (HIP error, allocation failure)."""


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so; a process that
    initialises two copies of the runtime sees the GPU through the first one only (the other one
    reports no device).  If torch is imported already its copy is in the process and libssdbox
    binds to it.  If torch is installed but not imported yet, its copy is loaded first, so that a
    later `import torch` finds the runtime it ships with already in place; without torch the
    system's /opt/rocm runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        locations = list(spec.submodule_search_locations) if spec is not None else []
    except (ImportError, ValueError, AttributeError):
        return
    for loc in locations:
        cand = os.path.join(loc, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libssdbox.so is not built (%s missing): run `make -C face_detection_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`; this package has no CPU "
            "fallback" % LIB_PATH)
    _preload_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error():
    msg = load().ssdbox_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc):
    """Maps a status code to the Python exception the drop-in layer promises."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise SsdboxError(msg)
