"""The entry point that allocates nothing (ssdbox_match_encode_static, scratch in the library's
fixed per-GPU array) against ssdbox_match_encode, called through the C ABI: every output
bit-identical, on two calls in a row of which the second is smaller, no device memory taken,
captured into a HIP graph and replayed, and sizes beyond the fixed array are rejected."""
import ctypes

import numpy as np
import pytest

from conftest import ssd_config
from face_detection_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _run(lib, fn, ctx, geom, A, boxes, labels, counts, size):
    import torch
    B, G = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    conf = torch.full((B, A), -7.0, dtype=torch.float32, device=dev)
    locs = torch.full((B, A, 4), -7.0, dtype=torch.float32, device=dev)
    midx = torch.full((B, A), -7, dtype=torch.int32, device=dev)
    miou = torch.full((B, A), -7.0, dtype=torch.float32, device=dev)
    rc = fn(ctx.handle, geom.handle, None, A, _p(boxes), _p(labels), _p(counts), B, G, float(size), 0.5,
            _p(conf), _p(locs), _p(midx), _p(miou))
    assert rc == 0, lib.ssdbox_last_error()
    ctx.synchronize()
    return [t.cpu().numpy() for t in (conf, locs, midx, miou)]


def test_static_scratch_entry_equals_allocating_entry():
    import torch
    from face_detection_amd import _lib
    from face_detection_amd.runtime import Context, geometry_for
    lib = _lib.load()
    torch.cuda.set_device(0)
    ctx = Context(0)
    # first a batch of 640x640 images with up to 300 boxes (more than one list of 256), then a smaller one (fewer images,
    # fewer boxes, smaller geometry): the array still holds the first call's contents
    for size, batch, gmax, seed in ((640, 8, 300, 3), (256, 3, 16, 4)):
        cfg = ssd_config(size)
        geom = geometry_for(cfg, 0)
        A = geom.num_anchors
        rng = np.random.default_rng(seed)
        boxes, labels, counts = S.make_ground_truth(rng, batch, size, gmax=gmax)
        labels[:, ::3] = 2.0                                  # labels are gathered, not assumed 1
        d = [torch.from_numpy(x).cuda() for x in (boxes, labels, counts)]
        want = _run(lib, lib.ssdbox_match_encode, ctx, geom, A, *d, size)
        got = _run(lib, lib.ssdbox_match_encode_static, ctx, geom, A, *d, size)
        for w, g in zip(want, got):
            assert w.dtype == g.dtype and w.shape == g.shape
            assert np.array_equal(w.view(np.uint32), g.view(np.uint32))
        assert (want[0] > 0).any() and not (want[2] == -7).any()
    # the same call twice in a row gives the same result (what must start at zero is cleared)
    again = _run(lib, lib.ssdbox_match_encode_static, ctx, geom, A, *d, size)
    for w, g in zip(want, again):
        assert np.array_equal(w.view(np.uint32), g.view(np.uint32))


def test_static_scratch_entry_allocates_nothing():
    """A context that has only ever served ssdbox_match_encode_static: after one warm call the
    device's free memory does not move over further calls (outputs allocated beforehand), and the
    results are the allocating entry's, computed on another context."""
    import torch
    from face_detection_amd import _lib
    from face_detection_amd.runtime import Context, geometry_for
    lib = _lib.load()
    torch.cuda.set_device(0)
    size, batch, gmax = 256, 6, 64
    cfg = ssd_config(size)
    geom = geometry_for(cfg, 0)
    A = geom.num_anchors
    rng = np.random.default_rng(12)
    boxes, labels, counts = S.make_ground_truth(rng, batch, size, gmax=gmax)
    d = [torch.from_numpy(x).cuda() for x in (boxes, labels, counts)]
    want = _run(lib, lib.ssdbox_match_encode, Context(0), geom, A, *d, size)
    fresh = Context(0, torch.cuda.Stream())                   # never sees the allocating entry
    conf = torch.empty((batch, A), dtype=torch.float32, device="cuda")
    locs = torch.empty((batch, A, 4), dtype=torch.float32, device="cuda")
    midx = torch.empty((batch, A), dtype=torch.int32, device="cuda")
    miou = torch.empty((batch, A), dtype=torch.float32, device="cuda")

    def call():
        rc = lib.ssdbox_match_encode_static(fresh.handle, geom.handle, None, A, _p(d[0]), _p(d[1]), _p(d[2]), batch,
                                            gmax, float(size), 0.5, _p(conf), _p(locs), _p(midx), _p(miou))
        assert rc == 0, lib.ssdbox_last_error()
    call()
    fresh.synchronize()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        call()
    fresh.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, "device memory moved by %d bytes over calls that must not allocate" % (free0 - free1)
    for w, g in zip(want, (conf, locs, midx, miou)):
        assert np.array_equal(w.view(np.uint32), g.cpu().numpy().view(np.uint32))


def test_static_scratch_entry_under_graph_capture():
    """The caller the entry point exists for: the call is captured into a HIP graph (where an
    allocation is an error), the graph is replayed on new inputs written into the captured
    buffers, and every replay gives the allocating entry's result."""
    import torch
    from face_detection_amd import _lib
    from face_detection_amd.runtime import Context, geometry_for
    lib = _lib.load()
    torch.cuda.set_device(0)
    size, batch, gmax = 256, 5, 48
    cfg = ssd_config(size)
    geom = geometry_for(cfg, 0)
    A = geom.num_anchors
    rng = np.random.default_rng(21)
    sets = [S.make_ground_truth(rng, batch, size, gmax=gmax) for _ in range(3)]
    d = [torch.from_numpy(x).cuda() for x in sets[0]]
    conf = torch.empty((batch, A), dtype=torch.float32, device="cuda")
    locs = torch.empty((batch, A, 4), dtype=torch.float32, device="cuda")
    stream = torch.cuda.Stream()
    ctx = Context(0, stream)

    def call():
        rc = lib.ssdbox_match_encode_static(ctx.handle, geom.handle, None, A, _p(d[0]), _p(d[1]), _p(d[2]), batch,
                                            gmax, float(size), 0.5, _p(conf), _p(locs), None, None)
        assert rc == 0, lib.ssdbox_last_error()
    with torch.cuda.stream(stream):
        call()                                                # warm: module load, the context's set-up
    stream.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream):
        call()
    ref = Context(0)
    for boxes, labels, counts in sets:
        for dst, src in zip(d, (boxes, labels, counts)):
            dst.copy_(torch.from_numpy(src))
        conf.fill_(-7.0)
        locs.fill_(-7.0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        w_conf = torch.empty_like(conf)
        w_locs = torch.empty_like(locs)
        rc = lib.ssdbox_match_encode(ref.handle, geom.handle, None, A, _p(d[0]), _p(d[1]), _p(d[2]), batch, gmax,
                                     float(size), 0.5, _p(w_conf), _p(w_locs), None, None)
        assert rc == 0, lib.ssdbox_last_error()
        ref.synchronize()
        assert np.array_equal(conf.cpu().numpy().view(np.uint32), w_conf.cpu().numpy().view(np.uint32))
        assert np.array_equal(locs.cpu().numpy().view(np.uint32), w_locs.cpu().numpy().view(np.uint32))
        assert (w_conf > 0).any()


def test_static_scratch_entry_rejects_what_does_not_fit():
    import torch
    from face_detection_amd import _lib
    from face_detection_amd.runtime import Context, geometry_for
    lib = _lib.load()
    torch.cuda.set_device(0)
    ctx = Context(0)
    size, batch, gmax = 64, 256, 4096                        # 256 * 4096 keys of 8 B = 8 MiB
    assert batch * gmax * 8 > _lib.STATIC_SCRATCH_BYTES
    geom = geometry_for(ssd_config(size), 0)
    A = geom.num_anchors
    boxes = torch.zeros((batch, gmax, 4), dtype=torch.float32, device="cuda")
    boxes[:, :, 2:] = 0.5
    labels = torch.ones((batch, gmax), dtype=torch.float32, device="cuda")
    counts = torch.ones((batch,), dtype=torch.int32, device="cuda")
    # outputs of full size: a call that was wrongly accepted would still write inside them
    conf = torch.full((batch, A), -7.0, dtype=torch.float32, device="cuda")
    locs = torch.full((batch, A, 4), -7.0, dtype=torch.float32, device="cuda")
    rc = lib.ssdbox_match_encode_static(ctx.handle, geom.handle, None, A, _p(boxes), _p(labels), _p(counts),
                                        batch, gmax, float(size), 0.5, _p(conf), _p(locs), None, None)
    assert rc == _lib.ERR_INVALID
    assert b"fixed array" in lib.ssdbox_last_error()
    ctx.synchronize()
    assert bool((conf == -7.0).all())                        # nothing was launched
