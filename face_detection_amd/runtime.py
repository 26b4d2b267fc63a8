"""Host-side plumbing over the C ABI: contexts, anchor geometry handles, buffer exchange.

PyTorch is used only for device memory, streams and DLPack exchange; every computation is
a libssdbox kernel.  Nothing here falls back to the CPU.
"""
import ctypes
import threading

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise RuntimeError("face_detection_amd needs an AMD MI355X (gfx950) GPU; none is visible "
                           "and there is no CPU fallback")
    return torch


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------------------ context
class Context:
    """One libssdbox context: a stream plus scratch buffers.  Not shareable between threads
    that call concurrently; default_context() hands out one per (thread, device, stream)."""

    def __init__(self, device=None, stream=None):
        torch = require_gpu()
        self.lib = _lib.load()
        self.device = torch.cuda.current_device() if device is None else int(device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.stream = stream
        handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
        out = ctypes.c_void_p()
        _lib.check(self.lib.ssdbox_ctx_create(self.device, ctypes.c_void_p(handle), 0, ctypes.byref(out)))
        self.handle = out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ssdbox_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _lib.check(self.lib.ssdbox_ctx_synchronize(self.handle))

    def set_timing(self, enabled):
        _lib.check(self.lib.ssdbox_ctx_set_timing(self.handle, int(bool(enabled))))

    def collect_timing(self):
        """{stage: (total ms, launches)} of everything recorded since the last collect."""
        ms = (ctypes.c_float * _lib.NUM_STAGES)()
        cnt = (ctypes.c_int32 * _lib.NUM_STAGES)()
        _lib.check(self.lib.ssdbox_ctx_collect_timing(self.handle, ms, cnt))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.STAGE_NAMES)}

    def selftest(self, x_log, expect_log, x_exp, expect_exp):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (x_log, expect_log, x_exp, expect_exp)]
        n = len(arrs[0])
        ptrs = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for a in arrs]
        _lib.check(self.lib.ssdbox_selftest(self.handle, *ptrs, n))


_tls = threading.local()


def default_context(device=None):
    torch = require_gpu()
    device = torch.cuda.current_device() if device is None else int(device)
    stream = torch.cuda.current_stream(device)
    key = (device, stream.cuda_stream)
    cache = getattr(_tls, "contexts", None)
    if cache is None:
        cache = _tls.contexts = {}
    ctx = cache.get(key)
    if ctx is None:
        ctx = cache[key] = Context(device, stream)
    return ctx


# ----------------------------------------------------------------------------- geometry
def _config_key(config):
    return (int(config['img_width']), int(config['img_height']),
            tuple(int(s) for s in config['steps']),
            tuple(float(s) for s in config['scales']),
            tuple(tuple(float(r) for r in rs) for rs in config['aspect_ratios_per_layer']),
            bool(config['two_boxes_for_ar1']), float(config['offsets']))


class Geometry:
    """Feature-map geometry of one SSD_CONFIG (reference config.py:52-69) on one device.
    device=-1 builds a host-only handle (layer table and anchor count, no kernels)."""

    def __init__(self, config, device):
        self.lib = _lib.load()
        self.key = _config_key(config)
        img_w, img_h, steps, scales, ratios, two_boxes, offsets = self.key
        if len(scales) != len(steps) + 1:
            raise ValueError("SSD config needs len(scales) == len(steps) + 1, got %d and %d"
                             % (len(scales), len(steps)))
        if len(ratios) != len(steps):
            raise ValueError("SSD config needs one aspect-ratio list per step")
        n = len(steps)
        c_steps = (ctypes.c_int32 * n)(*steps)
        c_scales = (ctypes.c_double * (n + 1))(*scales)
        c_counts = (ctypes.c_int32 * n)(*[len(r) for r in ratios])
        flat = [x for r in ratios for x in r]
        c_ratios = (ctypes.c_double * max(len(flat), 1))(*flat)
        out = ctypes.c_void_p()
        _lib.check(self.lib.ssdbox_geom_create(int(device), img_w, img_h, n, c_steps, c_scales,
                                               c_counts, c_ratios, int(two_boxes), offsets,
                                               ctypes.byref(out)))
        self.handle = out
        self.device = int(device)
        self.num_anchors = int(self.lib.ssdbox_geom_num_anchors(out))

    def layers(self):
        n = len(self.key[2])
        fw, fh, nb = ((ctypes.c_int32 * n)() for _ in range(3))
        self.lib.ssdbox_geom_layers(self.handle, fw, fh, nb)
        return [(fw[i], fh[i], nb[i]) for i in range(n)]

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.ssdbox_geom_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_geom_cache = {}
_geom_lock = threading.Lock()


def geometry_for(config, device):
    key = (_config_key(config), int(device))
    with _geom_lock:
        g = _geom_cache.get(key)
        if g is None:
            g = _geom_cache[key] = Geometry(config, device)
        return g


class AnchorGrid:
    """What generate_anchors(config) returns: an (A,4) float32 [cx, cy, w, h] array that is
    not stored.  Kernels given this object build anchors from the feature-map geometry on
    the fly; converting or indexing it materialises the array once, on the GPU."""

    def __init__(self, config, device=None):
        torch = require_gpu()
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.geometry = geometry_for(config, self.device)
        self.shape = (self.geometry.num_anchors, 4)
        self.dtype = np.dtype(np.float32)
        self.ndim = 2
        self._dev = None
        self._host = None
        self._lock = threading.Lock()

    def __len__(self):
        return self.shape[0]

    def tensor(self):
        """The materialised anchors as a CUDA torch tensor."""
        if self._dev is None:
            # The object is shared between threads and streams (the reference hands one anchor
            # array to six tf.data workers): one of them materialises it, and it is published
            # only when the kernel has finished, so that any stream may read it.
            with self._lock:
                if self._dev is None:
                    torch = _torch()
                    out = torch.empty(self.shape, dtype=torch.float32, device="cuda:%d" % self.device)
                    with torch.cuda.device(self.device):
                        ctx = default_context(self.device)
                        _lib.check(ctx.lib.ssdbox_generate_anchors(ctx.handle, self.geometry.handle, _ptr(out)))
                        ctx.synchronize()
                    self._dev = out
        return self._dev

    def numpy(self):
        if self._host is None:
            self._host = self.tensor().cpu().numpy()
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None and np.dtype(dtype) != a.dtype else a

    def __getitem__(self, item):
        return self.numpy()[item]

    def __dlpack__(self, *args, **kwargs):
        return self.tensor().__dlpack__(*args, **kwargs)

    def __dlpack_device__(self):
        return self.tensor().__dlpack_device__()


# ---------------------------------------------------------------------- buffer exchange
def _is_tf_tensor(x):
    mod = type(x).__module__ or ""
    return mod.startswith("tensorflow")


def to_device(x, dtype=None, device=None):
    """Returns (contiguous CUDA torch tensor, family) for numpy arrays, torch tensors,
    AnchorGrid, anything exporting __dlpack__ (tf.Tensor included) or array-likes.
    family is 'numpy', 'torch' or 'tf' and selects what from_device hands back."""
    torch = require_gpu()
    device = torch.cuda.current_device() if device is None else int(device)
    dev = "cuda:%d" % device
    tdtype = {None: None, np.float32: torch.float32, np.int32: torch.int32}[dtype]
    if isinstance(x, AnchorGrid):
        return x.tensor(), "numpy"
    if isinstance(x, torch.Tensor):
        family, t = "torch", x
    elif _is_tf_tensor(x):
        family = "tf"
        try:
            import tensorflow as tf
            t = torch.from_dlpack(tf.experimental.dlpack.to_dlpack(x))
        except Exception:
            t = torch.from_numpy(np.ascontiguousarray(x.numpy()))
    elif isinstance(x, np.ndarray):
        family, t = "numpy", torch.from_numpy(np.ascontiguousarray(x))
    elif hasattr(x, "__dlpack__"):
        family, t = "numpy", torch.from_dlpack(x)
    else:
        family, t = "numpy", torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if tdtype is not None and t.dtype != tdtype:
        t = t.to(tdtype)
    if not t.is_cuda or t.device.index != device:
        t = t.to(dev)
    return t.contiguous(), family


def from_device(t, family, like=None):
    if family == "torch":
        if like is not None and isinstance(like, _torch().Tensor) and not like.is_cuda:
            return t.cpu()
        return t
    if family == "tf":
        import tensorflow as tf
        return tf.convert_to_tensor(t.cpu().numpy())
    return t.cpu().numpy()
