#!/usr/bin/env python3
"""Benchmark of the SSD box pipeline on MI355X.

A step = one pass of the whole hot path over one batch of synthetic WiderFace-shaped input
resident in HBM: match+encode of the batch's ground truth (training side) followed by
decode + threshold + top-k + NMS of the batch's head outputs (inference side), called
through the C ABI.  Default workload: batch 32 @ 640x640, A = 102 000 anchors, up to 256
ground-truth boxes per image, score threshold 0.05 (BASELINE.json configs[2]).

  python bench.py --gpus N --steps K --warmup W

For N > 1 run under `python -m torch.distributed.run --nproc-per-node N`: every rank owns
its own batch (images are independent; no collective inside the timed loop), so scaling is
weak and `value` is the sum over ranks.  Rank 0 prints one JSON line.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable says otherwise).
# The bench leaves the setting alone and records it: with more than four queues active, the kernels
# of one queue start tens of microseconds after their predecessors (profiles/micro/chain_stamps.hip).


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--gmax", type=int, default=256)
    ap.add_argument("--ratios", default="config", choices=["config", "model"])
    ap.add_argument("--score-thr", type=float, default=0.05)
    ap.add_argument("--stress", action="store_true", help="256 ground-truth boxes in every image (the whole run)")
    ap.add_argument("--no-stress-leg", action="store_true",
                    help="skip the extra figures on a batch with 256 ground-truth boxes in every image")
    ap.add_argument("--streams", type=int, default=4,
                    help="batches in flight (one stream and context each); HIP maps streams onto "
                         "GPU_MAX_HW_QUEUES hardware queues (4 by default), more streams than queues "
                         "share them and measured slower")
    ap.add_argument("--input-sets", type=int, default=4,
                    help="synthetic batches (inputs and outputs in buffers of their own) the timed loops "
                         "rotate through; four are 313 MB of head outputs, more than the 256 MiB "
                         "memory-side cache")
    ap.add_argument("--stream-priority", type=int, default=-1,
                    help="priority of the batch streams: -1 high (default; measured 9 %% faster than "
                         "the best normal-priority setting), 0 normal")
    ap.add_argument("--stream-priority-list", default="",
                    help="comma-separated priorities, one per batch stream in turn (overrides --stream-priority)")
    ap.add_argument("--seeds", type=int, default=5,
                    help="synthetic batches the step is also timed on after the reported run (0: none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=24)
    ap.add_argument("--anchors", default="geometry", choices=["geometry", "array"],
                    help="anchors generated on the fly (default) or read from a materialised array")
    ap.add_argument("--profile-streams", type=int, default=1,
                    help="batches in flight under --profile-only (1: kernels back to back)")
    ap.add_argument("--profile-half", default="step", choices=["step", "match", "decode"],
                    help="what --profile-only runs: the whole step or one half")
    ap.add_argument("--profile-only", action="store_true",
                    help="few steps on one stream, no CPU baseline (for rocprofv3 runs)")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def make_inputs(args, rank, seed=0, stress=None, iset=0):
    """Synthetic batch of this rank as numpy arrays (no GPU involved).  `iset` numbers the input
    sets the timed loops rotate through: every set is a batch of its own."""
    import numpy as np
    from face_detection_amd import synthetic as S
    cfg = S.reference_ssd_config(args.size, args.ratios)
    rng = np.random.default_rng(1000 + rank + 100 * seed + 100003 * iset)
    gt_boxes, gt_labels, gt_counts = S.make_ground_truth(rng, args.batch, args.size, gmax=args.gmax,
                                                         stress=args.stress if stress is None else stress)
    y_pred = S.make_head_outputs(rng, cfg, gt_boxes, gt_counts)
    return cfg, gt_boxes, gt_labels, gt_counts, y_pred


class Workload:
    """Device-resident inputs, outputs and per-stream contexts for one rank.

    The timed loops rotate through `n_sets` input/output sets (at least four, one per batch in
    flight): every set is a synthetic batch of its own in buffers of its own, so consecutive
    steps never read the same head output.  Four sets are 4 x 78 MB of head outputs and
    4 x 65 MB of targets: more than the 256 MiB memory-side cache holds, so what a step reads
    comes from HBM as it would in a pipeline whose every batch is new."""

    def __init__(self, args, rank, inputs=None):
        import numpy as np
        import torch
        from face_detection_amd import _lib
        from face_detection_amd.runtime import Context, geometry_for
        self.torch, self.np, self.lib_mod = torch, np, _lib
        self.lib = _lib.load()
        self.args = args
        self.rank = rank
        self.dev = dev = torch.device("cuda", torch.cuda.current_device())
        self.n_sets = max(args.streams, args.input_sets, 1)
        self.sets = [None] * self.n_sets
        self.load_inputs(first=inputs)
        self.geom = geometry_for(self.cfg, dev.index)
        self.A = self.geom.num_anchors
        self.g_handle, self.a_ptr = self.geom.handle, None
        if args.anchors == "array":
            from face_detection_amd.runtime import AnchorGrid
            self.d_anchors = AnchorGrid(self.cfg, dev.index).tensor()
            self.g_handle, self.a_ptr = None, ctypes.c_void_p(self.d_anchors.data_ptr())
        B, A = args.batch, self.A
        self.slots = []
        prios = [int(x) for x in args.stream_priority_list.split(",") if x.strip()] or [args.stream_priority]
        for k in range(max(args.streams, 1)):
            stream = torch.cuda.Stream(device=dev, priority=prios[k % len(prios)])
            self.slots.append({"stream": stream, "ctx": Context(dev.index, stream),
                               "set": self.sets[len(self.slots) % self.n_sets]})
        for st in self.sets:
            st.update({
                "conf": torch.empty((B, A), dtype=torch.float32, device=dev),
                "locs": torch.empty((B, A, 4), dtype=torch.float32, device=dev),
                "ytrue": torch.empty((B, A, 6), dtype=torch.float32, device=dev),
                "dets": torch.empty((B, 1000, 6), dtype=torch.float32, device=dev),
                "keep": torch.empty((B, 1000), dtype=torch.int32, device=dev),
                "count": torch.empty((B,), dtype=torch.int32, device=dev),
            })
        torch.cuda.synchronize()
        self.warm_all()

    def warm_all(self):
        """Every slot once through every path the timed loops use, whatever --warmup says: a
        context allocates its scratch on first use, and no allocation or stream synchronisation
        of that kind may land in a timed region.  Also counts the kept boxes of every set."""
        thr = self.args.score_thr
        for slot in self.slots:
            st = self.sets[0]
            self.match(slot, st)
            self.match_ytrue(slot, st)
            self.args.score_thr = 0.0
            self.detect(slot, st)
            self.args.score_thr = thr
            self.detect(slot, st)
        for st in self.sets:
            self.detect(self.slots[0], st)
            self.torch.cuda.synchronize()
            st["n_keep"] = int(st["count"].sum().item())

    def load_inputs(self, seed=0, stress=None, first=None):
        """Puts one synthetic batch per set on the device (replacing the previous ones); `first`
        is set 0's batch when the caller has generated it already."""
        torch = self.torch
        torch.cuda.synchronize()
        for k in range(self.n_sets):
            inp = first if (k == 0 and first is not None) else \
                make_inputs(self.args, self.rank, seed, stress=stress, iset=k)
            self.cfg, gt_boxes, gt_labels, gt_counts, y_pred = inp
            st = self.sets[k] or {}
            st.update({
                "np": (gt_boxes, gt_labels, gt_counts, y_pred),
                "gt_total": int(gt_counts.sum()),
                "boxes": torch.from_numpy(gt_boxes).to(self.dev),
                "labels": torch.from_numpy(gt_labels).to(self.dev),
                "counts": torch.from_numpy(gt_counts).to(self.dev),
                "ypred": torch.from_numpy(y_pred).to(self.dev),
            })
            self.sets[k] = st
        torch.cuda.synchronize()

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    def match(self, slot, st=None):
        a = self.args
        st = st or slot["set"]
        rc = self.lib.ssdbox_match_encode(
            slot["ctx"].handle, self.g_handle, self.a_ptr, self.A, self._p(st["boxes"]),
            self._p(st["labels"]), self._p(st["counts"]), a.batch, a.gmax, float(a.size), 0.5,
            self._p(st["conf"]), self._p(st["locs"]), None, None)
        if rc != 0:
            self.lib_mod.check(rc)

    def match_ytrue(self, slot, st=None):
        """The same targets in the layout the training consumer takes (reference
        utils/datasets.py:155-159): rows [1 - pos, pos, locs]."""
        a = self.args
        st = st or slot["set"]
        rc = self.lib.ssdbox_match_encode_ytrue(
            slot["ctx"].handle, self.g_handle, self.a_ptr, self.A, self._p(st["boxes"]),
            self._p(st["labels"]), self._p(st["counts"]), a.batch, a.gmax, float(a.size), 0.5,
            self._p(st["ytrue"]))
        if rc != 0:
            self.lib_mod.check(rc)

    def detect(self, slot, st=None, ypred=None):
        a = self.args
        st = st or slot["set"]
        rc = self.lib.ssdbox_decode_nms(
            slot["ctx"].handle, self.g_handle, self.a_ptr, self.A,
            self._p(st["ypred"] if ypred is None else ypred), a.batch,
            a.score_thr, 0.45, 1000, None, self._p(st["dets"]), self._p(st["keep"]),
            self._p(st["count"]))
        if rc != 0:
            self.lib_mod.check(rc)

    def pick(self, i):
        return self.slots[i % len(self.slots)], self.sets[i % self.n_sets]

    def step(self, i):
        slot, st = self.pick(i)
        self.match(slot, st)
        self.detect(slot, st)

    def setup_loss(self):
        """Buffers for the loss figure: y_true of set 0 in the training layout (from the
        fused match kernel), loss, mined-negative mask and gradient."""
        torch, a = self.torch, self.args
        st = self.sets[0]
        B, A = a.batch, self.A
        self.match_ytrue(self.slots[0], st)
        self.loss_total = torch.empty((B,), dtype=torch.float32, device=self.dev)
        self.loss_grad = torch.empty((B, A, 6), dtype=torch.float32, device=self.dev)
        torch.cuda.synchronize()

    def loss(self, slot, with_grad=True):
        st = self.sets[0]
        rc = self.lib.ssdbox_ssd_loss(
            slot["ctx"].handle, self._p(st["ytrue"]), self._p(st["ypred"]), self.args.batch, self.A,
            3, 0, 1.0, self._p(self.loss_total), None, self._p(self.loss_grad) if with_grad else None)
        if rc != 0:
            self.lib_mod.check(rc)

    def algorithmic_bytes(self, st=None):
        """Per batch: match+encode reads 20 B per GT box and writes 20 B per anchor (24 B in the
        y_true layout); decode+NMS reads the 24 B head row of every anchor and writes 24 B per
        kept box plus a 4 B count per image.  Of one set, or the mean over the sets the loops
        rotate through."""
        B, A = self.args.batch, self.A
        sets = [st] if st is not None else self.sets
        n_keep = sum(s["n_keep"] for s in sets) / len(sets)
        gt = sum(s["gt_total"] for s in sets) / len(sets)
        return {"match": 20 * gt + 20 * A * B, "ytrue": 20 * gt + 24 * A * B,
                "detect": 24 * A * B + 24 * n_keep + 4 * B, "n_keep": n_keep, "gt_total": gt}


def timed_loop(torch, dist, world, fn, steps, warmup):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def stage_times(wl, steps):
    """Average device time per launch of every kernel stage, from HIP events the library
    attaches to each launch on the stream it runs on (hipExtLaunchKernel: the events take the
    kernel's own begin and end timestamps; one stream, so the numbers are not overlapped).
    The launches rotate through the input / output sets like the timed loops, so the streaming
    kernels read and write HBM, not the memory-side cache."""
    slot = wl.slots[0]
    ctx = slot["ctx"]
    for i in range(5):
        wl.match(slot, wl.sets[i % wl.n_sets])
        wl.detect(slot, wl.sets[i % wl.n_sets])
    ctx.synchronize()
    ctx.set_timing(True)
    for i in range(steps):
        wl.match(slot, wl.sets[i % wl.n_sets])
        wl.detect(slot, wl.sets[i % wl.n_sets])
    t = ctx.collect_timing()
    ctx.set_timing(False)
    return {k: (ms / n * 1e3) for k, (ms, n) in t.items() if n}       # microseconds


def stage_times_ytrue(wl, steps):
    """The match stage of the y_true layout alone (same protocol)."""
    slot = wl.slots[0]
    ctx = slot["ctx"]
    for i in range(3):
        wl.match_ytrue(slot, wl.sets[i % wl.n_sets])
    ctx.synchronize()
    ctx.set_timing(True)
    for i in range(steps):
        wl.match_ytrue(slot, wl.sets[i % wl.n_sets])
    t = ctx.collect_timing()
    ctx.set_timing(False)
    return {k: (ms / n * 1e3) for k, (ms, n) in t.items() if n}


_CPU = {}      # inputs of the CPU baseline, set before the worker pool forks


def _cpu_one(b, keep_outputs=False):
    """One image through the numpy restatement of the reference's TF-eager algorithm."""
    from oracle import box_oracle as O
    cfg, gt_boxes, gt_labels, gt_counts, y_pred, anchors, size, thr = _CPU["in"]
    g = int(gt_counts[b])
    t0 = time.perf_counter()
    conf, locs = O.compute_target(anchors, gt_boxes[b, :g], gt_labels[b, :g], float(size))
    t1 = time.perf_counter()
    O.generate_anchors(cfg)                               # inference.py:44 regenerates per image
    dets, keep = O.postprocess(y_pred[b], anchors, thr, 1.0, 1.0, 0.45, 1000)
    t2 = time.perf_counter()
    return (t1 - t0, t2 - t1, (conf, locs, dets, keep) if keep_outputs else None)


def cpu_baseline(args, inputs, n_images, workers=6, pools=True):
    """The numpy restatement of the reference's TF-eager algorithm (materialised (A,G) IoU,
    O(K^3) NMS loop, Python anchor loop) on the first images of this batch: one core (the
    reported value), then a pool of `workers` processes as in the reference's
    num_parallel_calls=6 (utils/datasets.py:42).  Runs before the GPU is touched (the pool
    forks).  Also returns the per-image outputs for the parity check of the GPU results."""
    import multiprocessing as mp
    from oracle import box_oracle as O
    cfg, gt_boxes, gt_labels, gt_counts, y_pred = inputs
    t0 = time.perf_counter()
    anchors = O.generate_anchors(cfg)                     # once per dataset on the training side
    t_anchor = time.perf_counter() - t0
    _CPU["in"] = (cfg, gt_boxes, gt_labels, gt_counts, y_pred, anchors, args.size, args.score_thr)
    n = min(n_images, args.batch)
    t_match = t_det = 0.0
    outputs = []
    for b in range(n):
        tm, td, out = _cpu_one(b, keep_outputs=True)
        t_match += tm
        t_det += td
        outputs.append(out)
    per_image = (t_match + t_det) / n
    # worker pools: six processes as in the reference, then every core this job may use (a
    # one-GPU box grants 16 CPUs however many the machine shows)
    all_cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16), 16))
    pool_rate = pool_all_rate = None
    try:
        if not pools:
            raise RuntimeError("reduced baseline (several ranks share the host)")
        with mp.get_context("fork").Pool(workers) as pool:
            t0 = time.perf_counter()
            pool.map(_cpu_one, range(n), chunksize=1)
            pool_rate = n / (time.perf_counter() - t0)
        if all_cores != workers:
            with mp.get_context("fork").Pool(all_cores) as pool:
                reps = list(range(n)) * max(1, (2 * all_cores) // n)
                t0 = time.perf_counter()
                pool.map(_cpu_one, reps, chunksize=1)
                pool_all_rate = len(reps) / (time.perf_counter() - t0)
    except Exception as e:                                # the figure is optional
        print("cpu pool baseline skipped: %r" % (e,), file=sys.stderr)
    # the scalar C restatement of the same algorithm (no (A,G) temporaries), one core
    c_rate = None
    try:
        from oracle import c_oracle as CO
        t0 = time.perf_counter()
        for b in range(n):
            g = int(gt_counts[b])
            CO.compute_target(anchors, gt_boxes[b, :g], gt_labels[b, :g], float(args.size))
            CO.postprocess(y_pred[b], anchors, args.score_thr, 1.0, 1.0, 0.45, 1000)
        c_rate = n / (time.perf_counter() - t0)
    except Exception as e:                                # the figure is optional
        print("C oracle baseline skipped: %r" % (e,), file=sys.stderr)
    res = {
        "value": 1.0 / per_image, "unit": "images/sec", "cores": 1, "kind": "port",
        "sample": "%d images of this batch (G=%s): numpy literal oracle, match+encode %.3f s/img, "
                  "anchors+decode+NMS %.3f s/img, one-off anchor generation %.2f s"
                  % (n, [int(x) for x in gt_counts[:n]], t_match / n, t_det / n, t_anchor),
        "match_encode_images_per_s": n / t_match, "decode_nms_images_per_s": n / t_det,
        "pool": {"workers": workers, "images_per_s": pool_rate},
        "pool_all_cores": {"workers": all_cores, "images_per_s": pool_all_rate},
        "c_restatement_images_per_s": c_rate,
    }
    return res, outputs


def parity_check(wl, outputs):
    """GPU results of set 0's batch against the CPU baseline's outputs for the same images:
    positives mask, kept indices and counts exact; floats within rtol 1e-5 / atol 1e-6 (the
    numpy oracle's exp/log differ from the kernels' by a few ULP).  Both output layouts of
    match+encode are checked."""
    np, torch = wl.np, wl.torch
    slot, st = wl.slots[0], wl.sets[0]
    wl.match(slot, st)
    wl.match_ytrue(slot, st)
    wl.detect(slot, st)
    torch.cuda.synchronize()
    n = len(outputs)
    conf = st["conf"][:n].cpu().numpy()
    locs = st["locs"][:n].cpu().numpy()
    ytrue = st["ytrue"][:n].cpu().numpy()
    dets = st["dets"][:n].cpu().numpy()
    keep = st["keep"][:n].cpu().numpy()
    count = st["count"][:n].cpu().numpy()
    ok_match = ok_nms = ok_float = ok_ytrue = 0
    for b, (c_conf, c_locs, c_dets, c_keep) in enumerate(outputs):
        k = len(c_keep)
        m_ok = np.array_equal(conf[b], c_conf)
        n_ok = int(count[b]) == k and np.array_equal(keep[b, :k], c_keep)
        f_ok = np.allclose(locs[b], c_locs, rtol=1e-5, atol=1e-6) and n_ok and \
            np.allclose(dets[b, :k], np.asarray(c_dets, np.float32).reshape(k, 6), rtol=1e-5, atol=1e-6)
        # y_true rows are [one_hot(conf, 2), locs] (reference utils/datasets.py:155-159): the same
        # kernel math as the (conf, locs) call, so bit-equal to it
        y_ok = np.array_equal(ytrue[b, :, 1], conf[b]) and np.array_equal(ytrue[b, :, 0], 1.0 - conf[b]) and \
            np.array_equal(ytrue[b, :, 2:].view(np.uint32), locs[b].view(np.uint32))
        ok_match += int(m_ok)
        ok_nms += int(n_ok)
        ok_float += int(f_ok)
        ok_ytrue += int(y_ok)
    return {"images": n, "match_mask_exact": ok_match, "nms_kept_exact": ok_nms, "floats_close": ok_float,
            "ytrue_equals_conf_locs": ok_ytrue}


def pcie_times(wl, reps=5):
    """Host<->device transfer time of one batch through pinned memory (inputs: head outputs
    and ground truth; outputs: targets and detections).  Reported next to `value`, which is
    measured with inputs already resident in HBM."""
    torch = wl.torch
    st = wl.sets[0]
    gt_boxes, gt_labels, gt_counts, y_pred = st["np"]
    h_in = [torch.from_numpy(x).pin_memory() for x in (y_pred, gt_boxes, gt_labels, gt_counts)]
    d_in = [st["ypred"], st["boxes"], st["labels"], st["counts"]]
    d_out = [st["conf"], st["locs"], st["dets"], st["keep"], st["count"]]
    h_out = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in d_out]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for h, d in zip(h_in, d_in):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
    h2d = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        for h, d in zip(h_out, d_out):
            h.copy_(d, non_blocking=True)
        torch.cuda.synchronize()
    d2h = (time.perf_counter() - t0) / reps
    nbytes_in = sum(h.numel() * h.element_size() for h in h_in)
    nbytes_out = sum(h.numel() * h.element_size() for h in h_out)
    return {"h2d_ms": h2d * 1e3, "d2h_ms": d2h * 1e3, "h2d_bytes": nbytes_in, "d2h_bytes": nbytes_out}


def single_image_latency(wl, reps=50):
    """Batch 1 (one 640x640 head output, as reference inference.py processes images): wall time
    of decode + threshold + top-k + NMS through the C ABI, input resident, nothing else in flight."""
    torch = wl.torch
    slot, st = wl.slots[0], wl.sets[0]
    a = wl.args

    def one():
        rc = wl.lib.ssdbox_decode_nms(
            slot["ctx"].handle, wl.g_handle, wl.a_ptr, wl.A, wl._p(st["ypred"]), 1, a.score_thr, 0.45, 1000,
            None, wl._p(st["dets"]), wl._p(st["keep"]), wl._p(st["count"]))
        if rc != 0:
            wl.lib_mod.check(rc)
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        one()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def hbm_copy_rate(torch, nbytes=1 << 28, reps=20):
    """Achievable HBM bandwidth on this box: device-to-device copy of 256 MiB, read + write
    bytes per second (the roofline fractions are quoted against the nominal peak instead)."""
    src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    dst = torch.empty_like(src)
    src.zero_()
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def single_batch_latency(wl, reps=30):
    """Wall time of one step on one stream with nothing else in flight."""
    torch = wl.torch
    slot = wl.slots[0]
    for i in range(3):
        wl.match(slot, wl.sets[i % wl.n_sets])
        wl.detect(slot, wl.sets[i % wl.n_sets])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        wl.match(slot, wl.sets[i % wl.n_sets])
        wl.detect(slot, wl.sets[i % wl.n_sets])
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


STEADY_STEPS = 2000        # timed steps of a steady-state region (after STEADY_WARMUP untimed ones)
STEADY_WARMUP = 200


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    # The CPU baseline forks a worker pool: it runs before anything initialises the GPU.  With
    # several ranks on one host rank 0 takes a reduced sample (fewer images, no worker pools)
    # while the others wait at the rendezvous, so that every line carries the baseline and the
    # parity gate.
    inputs = cpu = cpu_outputs = None
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    env_rank = int(os.environ.get("RANK", "0"))
    if env_rank == 0 and not args.no_cpu_baseline and not args.profile_only:
        inputs = make_inputs(args, 0)
        if env_world == 1:
            cpu, cpu_outputs = cpu_baseline(args, inputs, args.cpu_images)
        else:
            cpu, cpu_outputs = cpu_baseline(args, inputs, min(args.cpu_images, 4), pools=False)

    import torch
    import torch.distributed as dist
    from face_detection_amd import dist as fdist
    rank, world, _ = fdist.init_from_env()
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    backend = dist.get_backend() if world > 1 else None
    if world > 1 and not os.environ.get("SSDBOX_DIST_BACKEND"):
        # one process per GPU over RCCL: anything else is a rehearsal and has to say so
        assert backend == "nccl", "multi-GPU bench needs the nccl (RCCL) backend, got %r" % (backend,)
        assert world <= torch.cuda.device_count(), \
            "%d ranks but %d GPUs visible" % (world, torch.cuda.device_count())
    if args.profile_only:
        args.streams = args.profile_streams
    wl = Workload(args, rank, inputs)
    nslots = len(wl.slots)

    if args.profile_only:
        fn = {"step": wl.step, "match": lambda i: wl.match(*wl.pick(i)),
              "decode": lambda i: wl.detect(*wl.pick(i))}[args.profile_half]
        for i in range(args.warmup + args.steps):
            fn(i)
        torch.cuda.synchronize()
        if args.profile_half != "step":
            return
        for i in range(10):
            wl.match_ytrue(*wl.pick(i))
        torch.cuda.synchronize()
        wl.setup_loss()
        for _ in range(10):
            wl.loss(wl.slots[0])
        torch.cuda.synchronize()
        return

    # the region the caller asked for, exactly: W untimed steps, K timed ones
    elapsed_req = timed_loop(torch, dist, world, wl.step, args.steps, args.warmup)
    value_req = args.batch * args.steps * world / elapsed_req
    # A region of a few milliseconds reads the fill and drain of the chains in flight and the
    # first milliseconds after a pause, not the rate: `value` is then taken from a steady-state
    # region (STEADY_STEPS after STEADY_WARMUP untimed) and the requested one is reported next to it.
    if args.steps >= STEADY_STEPS:
        elapsed, timed_steps = elapsed_req, args.steps
    else:
        timed_steps = STEADY_STEPS
        elapsed = timed_loop(torch, dist, world, wl.step, timed_steps, STEADY_WARMUP)
    value = args.batch * timed_steps * world / elapsed

    # the two halves on their own (same protocol), for the per-half figures of the metric
    half_steps = max(args.steps, STEADY_STEPS)

    def half(fn):
        t = timed_loop(torch, dist, world, fn, half_steps, STEADY_WARMUP)
        return args.batch * half_steps * world / t
    r_match = half(lambda i: wl.match(*wl.pick(i)))
    r_ytrue = half(lambda i: wl.match_ytrue(*wl.pick(i)))
    r_det = half(lambda i: wl.detect(*wl.pick(i)))
    # every batch in flight reading ONE head-output buffer (78 MB, resident in the memory-side
    # cache): what the decode half reads when its input is not new every step
    shared = wl.sets[0]["ypred"]
    r_det_cached = half(lambda i: wl.detect(*wl.pick(i), ypred=shared))

    # evaluation.py:67 runs the same post-processing with threshold 0: every anchor is a candidate
    thr = args.score_thr
    args.score_thr = 0.0
    n_det0 = max(args.steps // 4, 10)
    t_det0 = timed_loop(torch, dist, world, lambda i: wl.detect(*wl.pick(i)), n_det0, 3)
    stages0 = stage_times(wl, 20) if rank == 0 else {}
    args.score_thr = thr

    # the training-side consumer of the targets: SSD loss with hard-negative mining, value and
    # gradient (not part of `value`)
    wl.setup_loss()
    t_loss = timed_loop(torch, dist, world, lambda i: wl.loss(wl.slots[0]), max(args.steps // 4, 10), 5)
    n_loss = max(args.steps // 4, 10)

    # the optional collective, outside the timed loop: every rank receives all detections
    allgather_ms = allgather_ok = None
    if world > 1:
        slot, st = wl.slots[0], wl.sets[0]
        gb = args.batch * world
        try:                                              # optional figure: never costs the line
            wl.detect(slot, st)
            for _ in range(3):
                fdist.all_gather_detections(st["dets"], st["keep"], st["count"], gb)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                fdist.all_gather_detections(st["dets"], st["keep"], st["count"], gb)
            torch.cuda.synchronize()
            allgather_ms = (time.perf_counter() - t0) / 10 * 1e3
            # what rank 0 received against what it computes itself from every rank's inputs
            g_dets, g_keep, g_count = fdist.all_gather_detections(st["dets"], st["keep"], st["count"], gb)
            if rank == 0:
                allgather_ok = True
                for r in range(world):
                    other = torch.from_numpy(make_inputs(args, r)[4]).to(wl.dev)
                    wl.detect(slot, st, ypred=other)
                    torch.cuda.synchronize()
                    sl = slice(r * args.batch, (r + 1) * args.batch)
                    allgather_ok = allgather_ok and bool(torch.equal(g_count[sl], st["count"])) and \
                        bool(torch.equal(g_keep[sl], st["keep"])) and \
                        bool(torch.equal(g_dets[sl].view(torch.int32), st["dets"].view(torch.int32)))
                wl.detect(slot, st)
                torch.cuda.synchronize()
        except Exception as e:
            print("rank %d: all-gather of detections failed: %r" % (rank, e), file=sys.stderr)

    # the same step on further synthetic batches (other seeds), after everything that describes
    # the reported batches: `value` stays the run above
    if rank == 0:
        stages = stage_times(wl, min(max(args.steps, 40), 100))
        stages_y = stage_times_ytrue(wl, 20)
        ab = wl.algorithmic_bytes()                       # mean over the sets: the timed loops
        ab0 = [wl.algorithmic_bytes(s) for s in wl.sets]  # per set: the per-kernel figures
        lat = single_batch_latency(wl)
        lat1 = single_image_latency(wl)
        pcie = pcie_times(wl)
        parity = parity_check(wl, cpu_outputs) if cpu is not None else None
    seed_steps = max(args.steps, 200)
    value_by_seed = [value]
    for seed in range(1, max(args.seeds, 1)):
        wl.load_inputs(seed)
        t = timed_loop(torch, dist, world, wl.step, seed_steps, 20)
        value_by_seed.append(args.batch * seed_steps * world / t)

    # the stress batches (256 ground-truth boxes in every image): the match stage alone and the step
    stress = None
    if not args.stress and not args.no_stress_leg:
        wl.load_inputs(0, stress=True)
        wl.warm_all()
        n_s = max(args.steps // 4, 100)
        t_s = timed_loop(torch, dist, world, wl.step, n_s, 10)
        t_sm = timed_loop(torch, dist, world, lambda i: wl.match(*wl.pick(i)), n_s, 10)
        if rank == 0:
            st = stage_times(wl, 20)
            sb = wl.algorithmic_bytes()["match"]
            stress = {"gt_total_per_batch": wl.algorithmic_bytes()["gt_total"], "match_us": st.get("match"),
                      "force_us": st.get("force"),
                      "frac": sb / (st["match"] * 1e-6) / 1e9 / HBM_PEAK_GBS if st.get("match") else None,
                      "step_images_per_s": args.batch * n_s * world / t_s,
                      "match_encode_images_per_s": args.batch * n_s * world / t_sm,
                      "match_encode_half_frac": sb / args.batch * (args.batch * n_s / t_sm) / 1e9 / HBM_PEAK_GBS}

    if rank == 0:
        KERNEL_SYMBOL = {"match": "match_geom_kernel", "force": "force_kernel", "select": "select_rows6_kernel",
                         "prune": "cand_prune_kernel", "topk": "topk_sort_kernel", "nms_mask": "nms_mask_kernel",
                         "nms_scan": "nms_scan_kernel"}
        # Every kernel against the bytes it moves itself (mean over the sets the launches rotate
        # through).  Only the two streaming kernels have a byte roofline; the others handle a few
        # hundred kilobytes per launch and are bound by latency (one workgroup per image, or a
        # launch a few microseconds long): their fraction is not a meaningful figure and is null.
        B, A = args.batch, wl.A
        mean = lambda key: sum(x[key] for x in ab0) / len(ab0)
        n_keep, gt_tot = mean("n_keep"), mean("gt_total")
        K = 1000
        own_bytes = {
            "match": mean("match"),
            "force": 28.0 * gt_tot + 20.0 * gt_tot,         # keys + boxes + labels read, forced rows written
            "select": 24.0 * A * B,                          # the head rows (candidate keys come on top)
            "topk": None, "prune": None,
            "nms_mask": 16.0 * K * B + K * (K / 2) / 8 * B, # ranked boxes read, lower-triangle bits written
            "nms_scan": K * (K / 2) / 8 * B + 24.0 * n_keep + 4.0 * B,
        }
        streaming = ("match", "select")
        per_kernel = {}
        for k, us in stages.items():
            if k not in own_bytes or not us:
                continue
            nb = own_bytes[k]
            gbs = nb / (us * 1e-6) / 1e9 if nb else None
            per_kernel[k] = {"us": round(us, 2), "bytes": nb and int(nb), "achieved_GBs": gbs,
                             "frac": (gbs / HBM_PEAK_GBS) if (gbs and k in streaming) else None,
                             "bound": "hbm" if k in streaming else "latency"}
        dominant = max(streaming, key=lambda k: stages.get(k, 0.0))       # the longest kernel of the step
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dominant)
            except Exception:
                traffic = None
        frac_of = lambda nbytes, rate: nbytes / args.batch * (rate / world) / 1e9 / HBM_PEAK_GBS
        out = {
            "metric": "images/sec (match+encode; decode+NMS) b=32 @640^2, % HBM roofline, 1-8 GPU",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / timed_steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "SSD+FPN box pipeline, match+encode then decode+NMS per batch, "
                            "batch %d @%dx%d, A=%d, <=%d GT/image%s, score_thr %.2f, nms 0.45, "
                            "limit 1000" % (args.batch, args.size, args.size, wl.A, args.gmax,
                                            " (all 256)" if args.stress else "", args.score_thr),
                "batch_per_gpu": args.batch, "anchors": wl.A, "gt_total_per_batch": gt_tot,
                "kept_per_batch": n_keep, "batches_in_flight": nslots, "input_sets": wl.n_sets,
                "input_bytes_in_rotation": int(wl.n_sets * 24 * A * B),
                "stream_priority": args.stream_priority, "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (HIP default)"),
                "parallelism": "image-sharded x%d" % world,
                "dist_backend": backend, "rccl_ranks": world if backend == "nccl" else 0,
            },
            # `value` is a steady-state region; the region --steps / --warmup ask for is next to it
            "timed_steps": timed_steps, "timed_warmup": STEADY_WARMUP if timed_steps != args.steps else args.warmup,
            "value_requested_region": value_req, "ms_per_step_requested_region": elapsed_req / args.steps * 1e3,
            "match_encode_images_per_s": r_match,
            "match_encode_ytrue_images_per_s": r_ytrue,
            "decode_nms_images_per_s": r_det,
            "decode_nms_images_per_s_cache_resident": r_det_cached,
            "half_steps": half_steps,
            "decode_nms_thr0_images_per_s": args.batch * n_det0 * world / t_det0,
            "kernel_us_thr0": {k: round(v, 2) for k, v in stages0.items()
                               if k in ("select", "prune", "topk", "nms_mask", "nms_scan")},
            "hbm_copy_GBs": hbm_copy_rate(torch),
            "ssd_loss_fwd_bwd_images_per_s": args.batch * n_loss * world / t_loss,
            "ssd_loss_fwd_bwd_ms": t_loss / n_loss * 1e3,
            "kernel_us": {k: round(v, 2) for k, v in stages.items()},
            "kernel_us_ytrue": {k: round(v, 2) for k, v in stages_y.items()},
            "roofline": {
                "bound": "hbm", "kernel": KERNEL_SYMBOL[dominant],
                "achieved": per_kernel[dominant]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": per_kernel[dominant]["frac"], "traffic": traffic,
                "algorithmic_bytes_per_launch": per_kernel[dominant]["bytes"],
                # the halves as they run in the timed loops (all their kernels, batches in flight):
                # algorithmic bytes of the half x its images per second / peak
                "match_encode_half_frac": frac_of(ab["match"], r_match),
                "match_encode_ytrue_half_frac": frac_of(ab["ytrue"], r_ytrue),
                "decode_nms_half_frac": frac_of(ab["detect"], r_det),
                "decode_nms_half_frac_cache_resident": frac_of(ab["detect"], r_det_cached),
                "ytrue_kernel_frac": (ab["ytrue"] / (stages_y["match"] * 1e-6) / 1e9 / HBM_PEAK_GBS)
                if stages_y.get("match") else None,
            },
            "roofline_kernels": per_kernel,
        }
        out["value_by_seed"] = value_by_seed
        out["value_median_of_seeds"] = sorted(value_by_seed)[len(value_by_seed) // 2]
        out["single_batch_latency_ms"] = lat * 1e3
        out["single_image_decode_nms_latency_ms"] = lat1 * 1e3
        if allgather_ms is not None:
            out["allgather_detections_ms"] = allgather_ms
            out["allgather_equals_local_recompute"] = allgather_ok
        step_s = elapsed / timed_steps
        pcie["images_per_s_with_transfers_serial"] = args.batch / (step_s + (pcie["h2d_ms"] + pcie["d2h_ms"]) * 1e-3)
        out["pcie"] = pcie
        if stress is not None:
            out["stress"] = stress
        parity_ok = True
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["parity"] = parity
            parity_ok = all(v == parity["images"] for k, v in parity.items() if k != "images")
            if not parity_ok:                             # a wrong result has no throughput
                out["value"] = None
        print(json.dumps(out))
        if not parity_ok:
            print("parity check against the CPU baseline failed: %r" % (parity,), file=sys.stderr)
            sys.exit(1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
