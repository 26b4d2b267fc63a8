"""The host layer without PyTorch: numpy in / numpy out through the library's own buffers, pinned
staging copies and DLPack exchange (ssdbox_buf_*, ssdbox_copy_*, ssdbox_dlpack_*), bit-exact
against the C oracle.  The drop-in calls run in a subprocess in which importing torch fails."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ssd_config
from oracle import box_oracle as O
from oracle import c_oracle as C
from face_detection_amd import synthetic as S

pytestmark = pytest.mark.gpu
F32 = np.float32
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_drop_in_calls_without_torch():
    code = textwrap.dedent("""
        import sys
        sys.modules['torch'] = None                      # any `import torch` now raises ImportError
        sys.path.insert(0, %r)
        sys.path.insert(0, %r)
        import numpy as np
        from conftest import ssd_config
        from oracle import c_oracle as C
        from face_detection_amd import synthetic as S
        import face_detection_amd as fda
        from face_detection_amd.utils import anchor_utils, box_utils

        def bits(a):
            return np.ascontiguousarray(a, np.float32).view(np.uint32)

        cfg = ssd_config(256)
        grid = anchor_utils.generate_anchors(cfg)
        anchors = C.generate_anchors(cfg)
        assert np.array_equal(bits(np.asarray(grid)), bits(anchors))
        rng = np.random.default_rng(11)
        boxes, labels, counts = S.make_ground_truth(rng, 3, 256, gmax=24)
        y = S.make_head_outputs(rng, cfg, boxes, counts, n_ties=16)
        for b in range(3):
            g = counts[b]
            # compute_target, decode, compute_nms: the reference's single-image calls
            for a in (grid, anchors):
                conf, locs = box_utils.compute_target(a, boxes[b, :g], labels[b, :g], 256.0)
                assert type(conf) is np.ndarray and type(locs) is np.ndarray
                c_conf, c_locs = C.compute_target(anchors, boxes[b, :g], labels[b, :g], 256.0)
                assert np.array_equal(conf, c_conf) and np.array_equal(bits(locs), bits(c_locs))
            dec = box_utils.decode(grid, y[b, :, 2:])
            assert np.array_equal(bits(dec), bits(C.decode(anchors, y[b, :, 2:])))
            mask = y[b, :, 1] > 0.05
            keep = box_utils.compute_nms(dec[mask], y[b, mask, 1], 0.45, 1000)
            assert keep.dtype == np.int32
            assert np.array_equal(keep, C.compute_nms(dec[mask], y[b, mask, 1], 0.45, 1000))
        # the batched calls, a 90 MB head output through the staging ring included
        conf, locs, midx, miou = fda.match_encode_batch(grid, boxes, counts, labels, 256.0, return_match=True)
        dets, keep, count = fda.decode_nms_batch(grid, y, 0.05, 0.45, 1000)
        for b in range(3):
            g = counts[b]
            c_conf, c_locs, c_idx, c_iou, _ = C.compute_target(anchors, boxes[b, :g], labels[b, :g], 256.0, details=True)
            assert np.array_equal(conf[b], c_conf) and np.array_equal(bits(locs[b]), bits(c_locs))
            assert np.array_equal(midx[b], c_idx) and np.array_equal(bits(miou[b]), bits(c_iou))
            c_dets, c_keep = C.postprocess(y[b], anchors, 0.05, 1.0, 1.0, 0.45, 1000)
            n = int(count[b])
            assert n == len(c_keep) and np.array_equal(keep[b, :n], c_keep)
            assert np.array_equal(bits(dets[b, :n]), bits(c_dets))
        # a context bound to a raw hipStream_t value (here: the stream of another context)
        from face_detection_amd.runtime import Context
        owner = Context(0, "own")
        raw = Context(0, owner.stream_handle())
        assert raw.stream_handle() == owner.stream_handle() and not raw.is_torch_stream
        d3, k3, c3 = fda.decode_nms_batch(grid, y, 0.05, 0.45, 1000, context=raw)
        assert np.array_equal(c3, count) and np.array_equal(k3, keep) and np.array_equal(bits(d3), bits(dets))
        t3 = fda.match_encode_batch(grid, boxes, counts, labels, 256.0, context=raw)
        assert np.array_equal(t3[0], conf) and np.array_equal(bits(t3[1]), bits(locs))
        big = ssd_config(640)
        gb = anchor_utils.generate_anchors(big)
        bb, lb, cb = S.make_ground_truth(rng, 2, 640, gmax=32)
        yb = np.concatenate([S.make_head_outputs(rng, big, bb, cb)] * 16)[:24]      # 24 x 102000 x 6 floats
        d2, k2, c2 = fda.decode_nms_batch(gb, yb, 0.05, 0.45, 1000)
        ab = C.generate_anchors(big)
        for b in (0, 1, 23):
            c_dets, c_keep = C.postprocess(yb[b], ab, 0.05, 1.0, 1.0, 0.45, 1000)
            n = int(c2[b])
            assert n == len(c_keep) and np.array_equal(k2[b, :n], c_keep) and np.array_equal(bits(d2[b, :n]), bits(c_dets))
        assert 'torch' not in [m for m, v in sys.modules.items() if v is not None and m == 'torch']
        print('torch-free ok')
    """) % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "torch-free ok" in out.stdout


def test_buffers_and_staged_copies():
    import ctypes
    from face_detection_amd.runtime import Context, DevArray
    ctx = Context(0, "own")
    rng = np.random.default_rng(3)
    for n in (0, 1, 1000, (4 << 20) // 4, (4 << 20) // 4 + 1, 5_000_003):       # around the 4 MiB staging chunk
        src = rng.standard_normal(n).astype(F32)
        d = DevArray.from_host(ctx, src)
        np.testing.assert_array_equal(d.numpy(), src)
        if n > 10:
            np.testing.assert_array_equal(d[3:9].numpy(), src[3:9])
            np.testing.assert_array_equal(d.reshape(1, n)[0].numpy(), src)
    # a freed buffer comes back for the next allocation of its size
    a = DevArray.empty(ctx, (1000,), F32)
    pa = a.data_ptr()
    del a
    b = DevArray.empty(ctx, (1000,), F32)
    assert b.data_ptr() == pa
    rc = ctx.lib.ssdbox_buf_free(ctx.handle, ctypes.c_void_p(12345))
    assert rc != 0 and b"not allocated" in ctx.lib.ssdbox_last_error()


def test_result_arrays_in_pinned_memory():
    """Results of the numpy path come back in pinned blocks from the library's pool (the copy from
    the device writes them directly): they are ordinary writable arrays, stay valid while held,
    and a block is reused only after every array over it is gone."""
    import ctypes
    import gc
    from face_detection_amd import runtime
    from face_detection_amd.runtime import Context, DevArray
    ctx = Context(0, "own")
    rng = np.random.default_rng(8)
    src = [rng.standard_normal((50_000, 4)).astype(F32) for _ in range(4)]
    dev = [DevArray.from_host(ctx, a) for a in src]
    held = [d.numpy() for d in dev]                       # four blocks of one size class, all held
    ptrs = {h.ctypes.data for h in held}
    assert len(ptrs) == 4
    for h, a in zip(held, src):
        np.testing.assert_array_equal(h, a)
        assert h.flags.writeable and h.flags.c_contiguous and h.base is not None
    row = held[0][123]                                     # a view keeps its block
    first = held[0].ctypes.data
    held[0] = None
    gc.collect()
    again = dev[1].numpy()
    assert again.ctypes.data not in ptrs or again.ctypes.data != first
    np.testing.assert_array_equal(row, src[0][123])
    del row
    gc.collect()
    reused = dev[2].numpy()                                # now the first block is free again
    assert reused.ctypes.data == first
    np.testing.assert_array_equal(reused, src[2])
    for h, a in zip(held[1:], src[1:]):
        np.testing.assert_array_equal(h, a)
    # small and very large results use ordinary memory; the C ABI refuses foreign pointers
    assert DevArray.from_host(ctx, np.arange(10, dtype=np.int32)).numpy().base is None
    rc = ctx.lib.ssdbox_host_free(ctypes.c_void_p(12345))
    assert rc != 0 and b"ssdbox_host_alloc" in ctx.lib.ssdbox_last_error()
    assert runtime._pinned_pool._outstanding >= 0


def test_dlpack_exchange():
    import torch
    from face_detection_amd.runtime import Context, DevArray
    ctx = Context(0, "own")
    rng = np.random.default_rng(4)
    src = rng.standard_normal((7, 5, 4)).astype(F32)
    # host tensor (numpy's __dlpack__, kDLCPU): staged into a buffer of the library's own
    d = DevArray.from_dlpack(ctx, src, src.shape)
    assert d._owned
    np.testing.assert_array_equal(d.numpy(), src)
    # GPU tensor (torch's __dlpack__, kDLROCM): used in place
    t = torch.from_numpy(src).cuda()
    torch.cuda.synchronize()
    z = DevArray.from_dlpack(ctx, t, tuple(t.shape))
    assert not z._owned and z.data_ptr() == t.data_ptr()
    np.testing.assert_array_equal(z.numpy(), src)
    with pytest.raises(ValueError):
        DevArray.from_dlpack(ctx, t.transpose(0, 1), (5, 7, 4))                  # not contiguous
    # A tensor whose producer is still running on another stream: the import orders the
    # context's stream behind it (through __dlpack__(stream=...) for an object, by waiting for
    # the device for a bare capsule), so the library never reads it half written.
    big = torch.empty((4096, 4096), dtype=torch.float32, device="cuda")
    want = torch.from_numpy(src).cuda()
    for as_capsule in (False, True):
        late = torch.zeros_like(want)
        for _ in range(30):
            big = big @ big * 1e-9                                                # milliseconds of queued work
        late.copy_(want)                                                         # ... then the value we expect
        obj = torch.utils.dlpack.to_dlpack(late) if as_capsule else late
        got = DevArray.from_dlpack(ctx, obj, tuple(late.shape)).numpy()
        np.testing.assert_array_equal(got, src)
    torch.cuda.synchronize()
    # export: the consumer owns the memory afterwards
    e = DevArray.from_host(ctx, src)
    back = torch.from_dlpack(e.to_dlpack())
    assert back.is_cuda and tuple(back.shape) == src.shape and back.dtype == torch.float32
    np.testing.assert_array_equal(back.cpu().numpy(), src)
    ints = DevArray.from_host(ctx, np.arange(12, dtype=np.int32).reshape(3, 4))
    np.testing.assert_array_equal(torch.from_dlpack(ints[1:].to_dlpack()).cpu().numpy(),
                                  np.arange(12, dtype=np.int32).reshape(3, 4)[1:])   # a view is copied first
    cap = DevArray.from_host(ctx, src).to_dlpack()
    del cap                                                                      # never consumed: freed by the capsule
    # objects that only speak DLPack go through the drop-in calls too
    from face_detection_amd.utils import box_utils

    class OnlyDLPack:
        def __init__(self, a):
            self._a, self.shape, self.dtype = a, a.shape, a.dtype

        def __dlpack__(self, *args, **kwargs):
            return self._a.__dlpack__()

        def __dlpack_device__(self):
            return self._a.__dlpack_device__()

    boxes = np.array([[0, 0, .5, .5], [.25, .25, .75, .75]], F32)
    got = box_utils.compute_iou(OnlyDLPack(boxes), OnlyDLPack(boxes))
    np.testing.assert_array_equal(got.view(np.uint32), O.compute_iou(boxes, boxes).view(np.uint32))
