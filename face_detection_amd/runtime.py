"""Host-side plumbing over the C ABI: contexts, anchor geometry handles, buffer exchange.

Every computation is a libssdbox kernel; nothing here falls back to the CPU.  Device memory,
host <-> device copies and DLPack exchange go through the library itself (ssdbox_buf_alloc,
ssdbox_copy_h2d / _d2h through a pinned staging ring, ssdbox_dlpack_import / _export), so numpy
arrays and tf.Tensors -- what the reference's callers pass, utils/datasets.py:130-152 and
inference.py:38-60 -- need no other GPU library.  PyTorch is used only when the caller passes
torch tensors (or binds a context to a torch stream): their memory and stream are then torch's.
"""
import ctypes
import sys
import threading

import numpy as np

from . import _lib


# ------------------------------------------------------------------------------ devices
_tls = threading.local()


def _torch_if_loaded():
    """The torch module when the caller has imported it already, else None (never imports it)."""
    return sys.modules.get("torch")


def _torch():
    import torch
    return torch


def device_count():
    return int(_lib.load().ssdbox_device_count())


def require_gpu():
    """Raises unless the library is built and an AMD GPU is visible; returns the library."""
    lib = _lib.load()
    if lib.ssdbox_device_count() <= 0:
        raise RuntimeError("face_detection_amd needs an AMD MI355X (gfx950) GPU; none is visible "
                           "and there is no CPU fallback")
    return lib


def set_device(index):
    """GPU used by this thread's calls that involve no torch tensor (default 0, or torch's
    current device when torch has initialised the GPU)."""
    _tls.device = int(index)


def _torch_runs_gpu():
    torch = _torch_if_loaded()
    try:
        return torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized()
    except Exception:
        return False


def current_device():
    dev = getattr(_tls, "device", None)
    if dev is not None:
        return dev
    if _torch_runs_gpu():
        return int(_torch().cuda.current_device())
    return 0


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ------------------------------------------------------------------------------ context
class Context:
    """One libssdbox context: a stream plus scratch buffers.  Not shareable between threads
    that call concurrently; default_context() hands out one per (thread, device, stream).
    stream: a torch.cuda.Stream, a raw hipStream_t value, or "own" for a non-blocking stream the
    context creates itself; None takes torch's current stream when torch is loaded and has
    initialised the GPU, else a stream of its own."""

    def __init__(self, device=None, stream=None):
        self.lib = require_gpu()
        self.device = current_device() if device is None else int(device)
        if stream is None:
            stream = _torch().cuda.current_stream(self.device) if _torch_runs_gpu() else "own"
        self.stream = stream
        out = ctypes.c_void_p()
        if isinstance(stream, str):
            _lib.check(self.lib.ssdbox_ctx_create(self.device, None, 1, ctypes.byref(out)))
        else:
            handle = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
            _lib.check(self.lib.ssdbox_ctx_create(self.device, ctypes.c_void_p(handle), 0, ctypes.byref(out)))
        self.handle = out

    @property
    def is_torch_stream(self):
        return hasattr(self.stream, "cuda_stream")

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

    def stream_handle(self):
        """The hipStream_t value the context launches on (0: the device's default stream)."""
        out = ctypes.c_void_p()
        _lib.check(self.lib.ssdbox_ctx_get_stream(self.handle, ctypes.byref(out)))
        return int(out.value or 0)

    def set_timing(self, enabled):
        _lib.check(self.lib.ssdbox_ctx_set_timing(self.handle, int(bool(enabled))))

    def collect_timing(self):
        """{stage: (total ms, launches)} of everything recorded since the last collect."""
        ms = (ctypes.c_float * _lib.NUM_STAGES)()
        cnt = (ctypes.c_int32 * _lib.NUM_STAGES)()
        _lib.check(self.lib.ssdbox_ctx_collect_timing(self.handle, ms, cnt))
        return {name: (float(ms[i]), int(cnt[i])) for i, name in enumerate(_lib.STAGE_NAMES)}

    def collect_timeline(self, ref=None, capacity=1 << 16):
        """[(stage name, begin ms, end ms)] of every launch recorded since the last collect_timing,
        on the time axis of `ref`'s first recorded launch (this context's when ref is None)."""
        stage = (ctypes.c_int32 * capacity)()
        t0 = (ctypes.c_float * capacity)()
        t1 = (ctypes.c_float * capacity)()
        n = ctypes.c_int32(0)
        _lib.check(self.lib.ssdbox_ctx_collect_timeline(self.handle, (ref or self).handle, stage, t0, t1,
                                                        capacity, ctypes.byref(n)))
        return [(_lib.STAGE_NAMES[stage[i]], float(t0[i]), float(t1[i])) for i in range(n.value)]

    def selftest(self, x_log, expect_log, x_exp, expect_exp):
        arrs = [np.ascontiguousarray(a, np.float32) for a in (x_log, expect_log, x_exp, expect_exp)]
        n = len(arrs[0])
        ptrs = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for a in arrs]
        _lib.check(self.lib.ssdbox_selftest(self.handle, *ptrs, n))


def default_context(device=None, torch_stream=None):
    """This thread's context for `device`: bound to torch's current stream when torch_stream is
    true (the call handles torch tensors), to a stream of the library's own when it is false,
    and to whichever applies when it is None (torch's if torch runs the GPU already)."""
    device = current_device() if device is None else int(device)
    if torch_stream is None:
        torch_stream = _torch_runs_gpu()
    cache = getattr(_tls, "contexts", None)
    if cache is None:
        cache = _tls.contexts = {}
    if torch_stream:
        stream = _torch().cuda.current_stream(device)
        key = (device, stream.cuda_stream)
    else:
        stream, key = "own", (device, "own")
    ctx = cache.get(key)
    if ctx is None:
        ctx = cache[key] = Context(device, stream)
    return ctx


# ----------------------------------------------------------------- library device arrays
_DL_CODE = {"i": 0, "u": 1, "f": 2}
_capsule_get = ctypes.pythonapi.PyCapsule_GetPointer
_capsule_get.restype, _capsule_get.argtypes = ctypes.c_void_p, [ctypes.py_object, ctypes.c_char_p]
_capsule_valid = ctypes.pythonapi.PyCapsule_IsValid
_capsule_valid.restype, _capsule_valid.argtypes = ctypes.c_int, [ctypes.py_object, ctypes.c_char_p]
_capsule_new = ctypes.pythonapi.PyCapsule_New
_capsule_new.restype, _capsule_new.argtypes = ctypes.py_object, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]


# the capsule destructor gets the object while it is being destroyed: it only handles its address
_raw_capsule_get = ctypes.PYFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p)(
    ("PyCapsule_GetPointer", ctypes.pythonapi))
_raw_capsule_valid = ctypes.PYFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_char_p)(
    ("PyCapsule_IsValid", ctypes.pythonapi))


@ctypes.CFUNCTYPE(None, ctypes.c_void_p)
def _capsule_destructor(capsule_addr):
    # a capsule nobody consumed still carries its name: free the tensor it holds
    try:
        if _raw_capsule_valid(capsule_addr, b"dltensor"):
            _lib.load().ssdbox_dlpack_delete(ctypes.c_void_p(_raw_capsule_get(capsule_addr, b"dltensor")))
    except Exception:
        pass


# ------------------------------------------------------------- pinned result arrays
class _PinnedBlock:
    """A pinned host block that numpy views as bytes; goes back to the pool when the last array
    over it is gone."""

    def __init__(self, pool, ptr, nbytes):
        self._pool, self._ptr, self._nbytes = pool, ptr, nbytes
        self.__array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self._pool.release(self._ptr, self._nbytes)
        except Exception:
            pass


class _PinnedPool:
    """Result arrays of the numpy path live in pinned host memory, so that the copy from the
    device writes them directly (ssdbox_copy_d2h needs no staging for them).  Blocks are pooled
    by size; large, tiny or too many outstanding results fall back to ordinary numpy memory."""
    GRANULE = 64 << 10
    MIN_BYTES, MAX_BYTES = 64 << 10, 64 << 20
    MAX_OUTSTANDING = 1 << 30          # pinned bytes in arrays the caller still holds
    MAX_CACHED = 256 << 20             # pinned bytes kept for reuse

    def __init__(self):
        self._lock = threading.Lock()
        self._free = {}                # size -> [ptr]
        self._cached = 0
        self._outstanding = 0

    def empty(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if not self.MIN_BYTES <= nbytes <= self.MAX_BYTES:
            return None
        size = -(-nbytes // self.GRANULE) * self.GRANULE
        ptr = None
        with self._lock:
            if self._outstanding + size > self.MAX_OUTSTANDING:
                return None
            blocks = self._free.get(size)
            if blocks:
                ptr = blocks.pop()
                self._cached -= size
            self._outstanding += size
        if ptr is None:
            out = ctypes.c_void_p()
            if _lib.load().ssdbox_host_alloc(size, ctypes.byref(out)) != 0 or not out.value:
                with self._lock:
                    self._outstanding -= size
                return None
            ptr = out.value
        raw = np.asarray(_PinnedBlock(self, ptr, size))
        return raw[:nbytes].view(dtype).reshape(shape)

    def release(self, ptr, size):
        with self._lock:
            self._outstanding -= size
            if self._cached + size <= self.MAX_CACHED:
                self._free.setdefault(size, []).append(ptr)
                self._cached += size
                return
        _lib.load().ssdbox_host_free(ctypes.c_void_p(ptr))


_pinned_pool = _PinnedPool()


class DevArray:
    """A dense row-major array in device memory that the library allocated (or borrowed through
    DLPack), with the few tensor operations the host layer needs.  Views (indexing the first
    axis, reshape) share the root's memory and keep it alive."""

    def __init__(self, ctx, ptr, shape, dtype, root=None, owned=True, keepalive=None):
        self.ctx, self._ptr = ctx, int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._root, self._owned, self._keepalive = root, owned and root is None, keepalive

    @classmethod
    def empty(cls, ctx, shape, dtype):
        shape = tuple(int(s) for s in shape)
        nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
        out = ctypes.c_void_p()
        _lib.check(ctx.lib.ssdbox_buf_alloc(ctx.handle, nbytes, ctypes.byref(out)))
        return cls(ctx, out.value, shape, dtype)

    @classmethod
    def from_host(cls, ctx, array, dtype=None):
        a = np.ascontiguousarray(array, dtype)
        d = cls.empty(ctx, a.shape, a.dtype)
        if a.nbytes:
            _lib.check(ctx.lib.ssdbox_copy_h2d(ctx.handle, ctypes.c_void_p(d._ptr), a.ctypes.data_as(ctypes.c_void_p),
                                               a.nbytes))
        return d

    @classmethod
    def from_dlpack(cls, ctx, obj, shape):
        """Borrows (GPU tensor) or stages (host tensor) what obj.__dlpack__() / a capsule describes.

        A GPU tensor is read in place on the context's stream.  An object with __dlpack__ is given
        that stream (the DLPack stream protocol: the producer orders its pending work before it);
        a bare capsule says nothing about its producer's stream, so the import waits for the
        device to be idle."""
        ordered = 0
        if type(obj).__name__ == "PyCapsule":
            capsule = obj
        else:
            capsule = None
            on_gpu = True
            try:
                on_gpu = int(obj.__dlpack_device__()[0]) != 1          # 1: kDLCPU
            except Exception:
                pass
            handle = ctx.stream_handle() if on_gpu else 0
            if handle:
                try:
                    capsule = obj.__dlpack__(stream=handle)
                    ordered = 1
                except (TypeError, RuntimeError, ValueError, NotImplementedError):
                    capsule = None                                     # a producer without the stream argument
            if capsule is None:
                capsule = obj.__dlpack__()
        managed = _capsule_get(capsule, b"dltensor")
        ptr, n = ctypes.c_void_p(), ctypes.c_int64()
        code, bits, owned = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _lib.check(ctx.lib.ssdbox_dlpack_import(ctx.handle, ctypes.c_void_p(managed), ordered, ctypes.byref(ptr),
                                                ctypes.byref(n), ctypes.byref(code), ctypes.byref(bits),
                                                ctypes.byref(owned)))
        kind = {0: "i", 1: "u", 2: "f"}.get(code.value)
        if kind is None or n.value != int(np.prod(shape, dtype=np.int64)):
            if owned.value:
                ctx.lib.ssdbox_buf_free(ctx.handle, ptr)
            raise ValueError("unsupported DLPack tensor (dtype code %d, %d elements)" % (code.value, n.value))
        dtype = np.dtype("%s%d" % (kind, bits.value // 8))
        # the capsule stays unconsumed: its producer frees the tensor when the capsule goes
        return cls(ctx, ptr.value or 0, shape, dtype, owned=bool(owned.value), keepalive=(capsule, obj))

    # ---- the tensor operations the host layer uses
    def data_ptr(self):
        return self._ptr

    @property
    def ndim(self):
        return len(self.shape)

    def dim(self):
        return len(self.shape)

    def numel(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.numel() * self.dtype.itemsize

    def contiguous(self):
        return self

    def _view(self, ptr, shape):
        return DevArray(self.ctx, ptr, shape, self.dtype, root=self._root or self)

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], int) else shape
        if int(np.prod(shape, dtype=np.int64)) != self.numel():
            raise ValueError("cannot reshape %s to %s" % (self.shape, tuple(shape)))
        return self._view(self._ptr, shape)

    def __getitem__(self, item):
        if item is None:
            return self._view(self._ptr, (1,) + self.shape)
        if not self.shape:
            raise IndexError("index into a 0-d array")
        row = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        if isinstance(item, slice):
            start, stop, step = item.indices(self.shape[0])
            if step != 1:
                raise IndexError("only contiguous slices of the first axis")
            return self._view(self._ptr + start * row, (max(stop - start, 0),) + self.shape[1:])
        i = int(item)
        if not -self.shape[0] <= i < self.shape[0]:
            raise IndexError(i)
        return self._view(self._ptr + (i % self.shape[0]) * row, self.shape[1:])

    def numpy(self):
        """A host copy; returns when the data has arrived (and the work before it has finished)."""
        out = _pinned_pool.empty(self.shape, self.dtype)
        if out is None:
            out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            _lib.check(self.ctx.lib.ssdbox_copy_d2h(self.ctx.handle, out.ctypes.data_as(ctypes.c_void_p),
                                                    ctypes.c_void_p(self._ptr), out.nbytes))
        return out

    def clone(self):
        out = DevArray.empty(self.ctx, self.shape, self.dtype)
        if self.nbytes:
            _lib.check(self.ctx.lib.ssdbox_copy_d2d(self.ctx.handle, ctypes.c_void_p(out._ptr),
                                                    ctypes.c_void_p(self._ptr), self.nbytes))
        return out

    def to_dlpack(self):
        """A "dltensor" capsule that owns the memory from now on (views are copied first)."""
        src = self if (self._root is None and self._owned) else self.clone()
        shape = (ctypes.c_int64 * max(len(src.shape), 1))(*src.shape)
        managed = ctypes.c_void_p()
        _lib.check(src.ctx.lib.ssdbox_dlpack_export(src.ctx.handle, ctypes.c_void_p(src._ptr), len(src.shape), shape,
                                                    _DL_CODE[src.dtype.kind], src.dtype.itemsize * 8,
                                                    ctypes.byref(managed)))
        src._owned = False                   # the tensor's deleter frees it
        return _capsule_new(managed, b"dltensor", ctypes.cast(_capsule_destructor, ctypes.c_void_p))

    def __del__(self):
        try:
            if self._owned and self._ptr and getattr(self.ctx, "handle", None):
                self.ctx.lib.ssdbox_buf_free(self.ctx.handle, ctypes.c_void_p(self._ptr))
                self._owned = False
        except Exception:
            pass


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
        require_gpu()
        self.device = current_device() if device is None else int(device)
        self.geometry = geometry_for(config, self.device)
        self.shape = (self.geometry.num_anchors, 4)
        self.dtype = np.dtype(np.float32)
        self.ndim = 2
        self._dev = None
        self._host = None
        self._torch = None
        self._lock = threading.Lock()

    def __len__(self):
        return self.shape[0]

    def device_array(self):
        """The materialised anchors in library memory (DevArray)."""
        if self._dev is None:
            # The object is shared between threads and streams (the reference hands one anchor
            # array to six tf.data workers): one of them materialises it, and it is published
            # only when the kernel has finished, so that any stream may read it.
            with self._lock:
                if self._dev is None:
                    ctx = default_context(self.device, torch_stream=False)
                    out = DevArray.empty(ctx, self.shape, np.float32)
                    _lib.check(ctx.lib.ssdbox_generate_anchors(ctx.handle, self.geometry.handle, _ptr(out)))
                    ctx.synchronize()
                    self._dev = out
        return self._dev

    def tensor(self):
        """The materialised anchors as a CUDA torch tensor (a copy in torch's memory)."""
        if self._torch is None:
            torch = _torch()
            host = torch.from_numpy(self.numpy())
            with self._lock:
                if self._torch is None:
                    t = host.to("cuda:%d" % self.device)
                    torch.cuda.current_stream(self.device).synchronize()
                    self._torch = t
        return self._torch

    def numpy(self):
        if self._host is None:
            self._host = self.device_array().numpy()
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


def _is_torch_tensor(x):
    torch = _torch_if_loaded()
    return torch is not None and isinstance(x, torch.Tensor)


class Session:
    """One call's view of the device.  If any argument is a torch tensor (or the caller's context
    is bound to a torch stream) arrays are torch CUDA tensors on torch's current stream; otherwise
    they are DevArrays in library memory on a stream of the library's own, and torch is never
    imported."""

    def __init__(self, *args, context=None, device=None):
        require_gpu()
        self.use_torch = any(_is_torch_tensor(a) for a in args) or (context is not None and context.is_torch_stream)
        if context is not None:
            if self.use_torch and not context.is_torch_stream:
                raise ValueError("torch tensors need a context bound to a torch.cuda.Stream (this one runs on "
                                 "a stream of its own or on a raw handle)")
            self.ctx = context
            self.device = context.device
        else:
            if self.use_torch:
                self.device = int(_torch().cuda.current_device()) if device is None else int(device)
            else:
                self.device = current_device() if device is None else int(device)
            self.ctx = default_context(self.device, torch_stream=self.use_torch)
        self.lib = self.ctx.lib

    # ---- inputs
    def put(self, x, dtype=None):
        """(device array, family): family is 'numpy', 'torch' or 'tf' and selects what get()
        hands back."""
        np_dtype = np.dtype(dtype) if dtype is not None else None
        if isinstance(x, AnchorGrid):
            return (x.tensor() if self.use_torch else x.device_array()), "numpy"
        if isinstance(x, DevArray):
            return x, "numpy"
        if self.use_torch:
            return self._put_torch(x, np_dtype)
        if _is_tf_tensor(x):
            import tensorflow as tf
            if np_dtype is not None and x.dtype.as_numpy_dtype != np_dtype:
                x = tf.cast(x, np_dtype)
            try:
                return DevArray.from_dlpack(self.ctx, tf.experimental.dlpack.to_dlpack(x), tuple(x.shape)), "tf"
            except Exception:
                return DevArray.from_host(self.ctx, x.numpy(), np_dtype), "tf"
        if isinstance(x, np.ndarray):
            return DevArray.from_host(self.ctx, x, np_dtype), "numpy"
        if hasattr(x, "__dlpack__") and hasattr(x, "shape"):
            try:
                same = np_dtype is None or np.dtype(getattr(x, "dtype", None)) == np_dtype
            except TypeError:
                same = False
            if same:
                try:
                    return DevArray.from_dlpack(self.ctx, x, tuple(x.shape)), "numpy"
                except (ValueError, NotImplementedError, RuntimeError, TypeError, BufferError):
                    pass
        return DevArray.from_host(self.ctx, np.asarray(x), np_dtype), "numpy"

    def _put_torch(self, x, np_dtype):
        torch = _torch()
        dev = "cuda:%d" % self.device
        tdtype = {None: None, np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32}[np_dtype]
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
        if not t.is_cuda or t.device.index != self.device:
            t = t.to(dev)
        return t.contiguous(), family

    # ---- scratch and outputs
    def empty(self, shape, dtype=np.float32):
        if self.use_torch:
            torch = _torch()
            tdtype = {np.dtype(np.float32): torch.float32, np.dtype(np.int32): torch.int32,
                      np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
            return torch.empty(tuple(int(s) for s in shape), dtype=tdtype, device="cuda:%d" % self.device)
        return DevArray.empty(self.ctx, shape, dtype)

    def from_host(self, array):
        """A small host array to the device (counts, constants)."""
        if self.use_torch:
            return _torch().from_numpy(np.ascontiguousarray(array)).to("cuda:%d" % self.device)
        return DevArray.from_host(self.ctx, array)

    def host(self, arr):
        """numpy copy of a device array of this session."""
        if isinstance(arr, DevArray):
            return arr.numpy()
        return arr.cpu().numpy()

    def get(self, arr, family, like=None):
        """The result in the caller's family."""
        if family == "torch" and not isinstance(arr, DevArray):
            if like is not None and _is_torch_tensor(like) and not like.is_cuda:
                return arr.cpu()
            return arr
        if family == "tf":
            import tensorflow as tf
            if isinstance(arr, DevArray):
                # the capsule owns the memory from here on; while it is alive and unconsumed the
                # memory is still valid, so the host copy of the fallback is taken before it goes
                src = arr if (arr._root is None and arr._owned) else arr.clone()
                capsule = src.to_dlpack()
                try:
                    return tf.experimental.dlpack.from_dlpack(capsule)
                except Exception:
                    if not _capsule_valid(capsule, b"dltensor"):
                        raise                    # consumed before it failed: the memory is TensorFlow's
                    host = src.numpy()
                    del capsule
                    return tf.convert_to_tensor(host)
            return tf.convert_to_tensor(arr.cpu().numpy())
        return self.host(arr)
