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

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); two batch streams
# that share one run one after the other.  Eight queues leave each of the four batch streams one of
# its own next to the streams PyTorch creates (measured: 9.2e5 images/s against 8.9e5).  Must be set
# before the HIP runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


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
                    help="batches in flight (one stream and context each); HIP runs 4 hardware queues per "
                         "process, more streams share them and measured slower")
    ap.add_argument("--stream-priority", type=int, default=-1,
                    help="priority of the batch streams: -1 high (default; measured 9 %% faster than "
                         "the best normal-priority setting), 0 normal")
    ap.add_argument("--seeds", type=int, default=5,
                    help="synthetic batches the step is also timed on after the reported run (0: none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=24)
    ap.add_argument("--anchors", default="geometry", choices=["geometry", "array"],
                    help="anchors generated on the fly (default) or read from a materialised array")
    ap.add_argument("--profile-streams", type=int, default=1,
                    help="batches in flight under --profile-only (1: kernels back to back)")
    ap.add_argument("--profile-only", action="store_true",
                    help="few steps on one stream, no CPU baseline (for rocprofv3 runs)")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", os.environ.get("MASTER_PORT", "29511"), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def make_inputs(args, rank, seed=0, stress=None):
    """Synthetic batch of this rank as numpy arrays (no GPU involved)."""
    import numpy as np
    from face_detection_amd import synthetic as S
    cfg = S.reference_ssd_config(args.size, args.ratios)
    rng = np.random.default_rng(1000 + rank + 100 * seed)
    gt_boxes, gt_labels, gt_counts = S.make_ground_truth(rng, args.batch, args.size, gmax=args.gmax,
                                                         stress=args.stress if stress is None else stress)
    y_pred = S.make_head_outputs(rng, cfg, gt_boxes, gt_counts)
    return cfg, gt_boxes, gt_labels, gt_counts, y_pred


class Workload:
    """Device-resident inputs and per-stream outputs/contexts for one rank."""

    def __init__(self, args, rank, inputs=None):
        import numpy as np
        import torch
        from face_detection_amd import _lib
        from face_detection_amd.runtime import Context, geometry_for
        self.torch, self.np, self.lib_mod = torch, np, _lib
        self.lib = _lib.load()
        self.args = args
        dev = torch.device("cuda", torch.cuda.current_device())
        self.load_inputs(inputs or make_inputs(args, rank), dev)
        self.geom = geometry_for(self.cfg, dev.index)
        self.A = self.geom.num_anchors
        self.g_handle, self.a_ptr = self.geom.handle, None
        if args.anchors == "array":
            from face_detection_amd.runtime import AnchorGrid
            self.d_anchors = AnchorGrid(self.cfg, dev.index).tensor()
            self.g_handle, self.a_ptr = None, ctypes.c_void_p(self.d_anchors.data_ptr())
        B, A = args.batch, self.A
        self.slots = []
        for _ in range(max(args.streams, 1)):
            stream = torch.cuda.Stream(device=dev, priority=args.stream_priority)
            self.slots.append({
                "stream": stream,
                "ctx": Context(dev.index, stream),
                "conf": torch.empty((B, A), dtype=torch.float32, device=dev),
                "locs": torch.empty((B, A, 4), dtype=torch.float32, device=dev),
                "dets": torch.empty((B, 1000, 6), dtype=torch.float32, device=dev),
                "keep": torch.empty((B, 1000), dtype=torch.int32, device=dev),
                "count": torch.empty((B,), dtype=torch.int32, device=dev),
            })
        torch.cuda.synchronize()
        self.warm_all()

    def warm_all(self):
        """Every slot once through every path the timed loops use, whatever --warmup says: a
        context allocates its scratch on first use, and no allocation or stream synchronisation
        of that kind may land in a timed region."""
        thr = self.args.score_thr
        for slot in self.slots:
            self.match(slot)
            self.detect(slot)
            self.args.score_thr = 0.0
            self.detect(slot)
            self.args.score_thr = thr
        self.torch.cuda.synchronize()

    def load_inputs(self, inputs, dev=None):
        """Puts one synthetic batch on the device (replacing the previous one)."""
        torch = self.torch
        dev = dev or self.d_ypred.device
        torch.cuda.synchronize()
        self.cfg, self.gt_boxes, self.gt_labels, self.gt_counts, self.y_pred = inputs
        self.d_boxes = torch.from_numpy(self.gt_boxes).to(dev)
        self.d_labels = torch.from_numpy(self.gt_labels).to(dev)
        self.d_counts = torch.from_numpy(self.gt_counts).to(dev)
        self.d_ypred = torch.from_numpy(self.y_pred).to(dev)
        torch.cuda.synchronize()

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr())

    def match(self, slot):
        a = self.args
        rc = self.lib.ssdbox_match_encode(
            slot["ctx"].handle, self.g_handle, self.a_ptr, self.A, self._p(self.d_boxes),
            self._p(self.d_labels), self._p(self.d_counts), a.batch, a.gmax, float(a.size), 0.5,
            self._p(slot["conf"]), self._p(slot["locs"]), None, None)
        if rc != 0:
            self.lib_mod.check(rc)

    def detect(self, slot):
        a = self.args
        rc = self.lib.ssdbox_decode_nms(
            slot["ctx"].handle, self.g_handle, self.a_ptr, self.A, self._p(self.d_ypred), a.batch,
            a.score_thr, 0.45, 1000, None, self._p(slot["dets"]), self._p(slot["keep"]),
            self._p(slot["count"]))
        if rc != 0:
            self.lib_mod.check(rc)

    def step(self, i):
        slot = self.slots[i % len(self.slots)]
        self.match(slot)
        self.detect(slot)

    def setup_loss(self):
        """Buffers for the loss figure: y_true of this batch in the training layout (from the
        fused match kernel), loss, mined-negative mask and gradient."""
        torch, a = self.torch, self.args
        slot = self.slots[0]
        B, A = a.batch, self.A
        self.y_true = torch.empty((B, A, 6), dtype=torch.float32, device=slot["conf"].device)
        rc = self.lib.ssdbox_match_encode_ytrue(
            slot["ctx"].handle, self.g_handle, self.a_ptr, self.A, self._p(self.d_boxes),
            self._p(self.d_labels), self._p(self.d_counts), a.batch, a.gmax, float(a.size), 0.5,
            self._p(self.y_true))
        if rc != 0:
            self.lib_mod.check(rc)
        self.loss_total = torch.empty((B,), dtype=torch.float32, device=self.y_true.device)
        self.loss_grad = torch.empty((B, A, 6), dtype=torch.float32, device=self.y_true.device)
        torch.cuda.synchronize()

    def loss(self, slot, with_grad=True):
        rc = self.lib.ssdbox_ssd_loss(
            slot["ctx"].handle, self._p(self.y_true), self._p(self.d_ypred), self.args.batch, self.A,
            3, 0, 1.0, self._p(self.loss_total), None, self._p(self.loss_grad) if with_grad else None)
        if rc != 0:
            self.lib_mod.check(rc)

    def algorithmic_bytes(self):
        """Per batch: match+encode reads 20 B per GT box and writes 20 B per anchor;
        decode+NMS reads the 24 B head row of every anchor and writes 24 B per kept box
        plus a 4 B count per image."""
        B, A = self.args.batch, self.A
        n_keep = int(self.slots[0]["count"].sum().item())
        match_bytes = 20 * int(self.gt_counts.sum()) + 20 * A * B
        detect_bytes = 24 * A * B + 24 * n_keep + 4 * B
        return match_bytes, detect_bytes, n_keep


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
    kernel's own begin and end timestamps; one stream, so the numbers are not overlapped)."""
    slot = wl.slots[0]
    ctx = slot["ctx"]
    for _ in range(5):
        wl.match(slot)
        wl.detect(slot)
    ctx.synchronize()
    ctx.set_timing(True)
    for _ in range(steps):
        wl.match(slot)
        wl.detect(slot)
    t = ctx.collect_timing()
    ctx.set_timing(False)
    return {k: (ms / n * 1e3) for k, (ms, n) in t.items() if n}       # microseconds


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


def cpu_baseline(args, inputs, n_images, workers=6):
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
    """GPU results of the bench batch against the CPU baseline's outputs for the same images:
    positives mask, kept indices and counts exact; floats within rtol 1e-5 / atol 1e-6 (the
    numpy oracle's exp/log differ from the kernels' by a few ULP)."""
    np, torch = wl.np, wl.torch
    slot = wl.slots[0]
    wl.match(slot)
    wl.detect(slot)
    torch.cuda.synchronize()
    n = len(outputs)
    conf = slot["conf"][:n].cpu().numpy()
    locs = slot["locs"][:n].cpu().numpy()
    dets = slot["dets"][:n].cpu().numpy()
    keep = slot["keep"][:n].cpu().numpy()
    count = slot["count"][:n].cpu().numpy()
    ok_match = ok_nms = ok_float = 0
    for b, (c_conf, c_locs, c_dets, c_keep) in enumerate(outputs):
        k = len(c_keep)
        m_ok = np.array_equal(conf[b], c_conf)
        n_ok = int(count[b]) == k and np.array_equal(keep[b, :k], c_keep)
        f_ok = np.allclose(locs[b], c_locs, rtol=1e-5, atol=1e-6) and n_ok and \
            np.allclose(dets[b, :k], np.asarray(c_dets, np.float32).reshape(k, 6), rtol=1e-5, atol=1e-6)
        ok_match += int(m_ok)
        ok_nms += int(n_ok)
        ok_float += int(f_ok)
    return {"images": n, "match_mask_exact": ok_match, "nms_kept_exact": ok_nms, "floats_close": ok_float}


def pcie_times(wl, reps=5):
    """Host<->device transfer time of one batch through pinned memory (inputs: head outputs
    and ground truth; outputs: targets and detections).  Reported next to `value`, which is
    measured with inputs already resident in HBM."""
    torch = wl.torch
    slot = wl.slots[0]
    h_in = [torch.from_numpy(x).pin_memory() for x in (wl.y_pred, wl.gt_boxes, wl.gt_labels, wl.gt_counts)]
    d_in = [wl.d_ypred, wl.d_boxes, wl.d_labels, wl.d_counts]
    d_out = [slot["conf"], slot["locs"], slot["dets"], slot["keep"], slot["count"]]
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
    slot = wl.slots[0]
    a = wl.args

    def one():
        rc = wl.lib.ssdbox_decode_nms(
            slot["ctx"].handle, wl.g_handle, wl.a_ptr, wl.A, wl._p(wl.d_ypred), 1, a.score_thr, 0.45, 1000,
            None, wl._p(slot["dets"]), wl._p(slot["keep"]), wl._p(slot["count"]))
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
    for _ in range(3):
        wl.match(slot)
        wl.detect(slot)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        wl.match(slot)
        wl.detect(slot)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))

    # the CPU baseline forks a worker pool: it runs before anything initialises the GPU
    inputs = cpu = cpu_outputs = None
    single = int(os.environ.get("WORLD_SIZE", "1")) == 1
    if single and not args.no_cpu_baseline and not args.profile_only:
        inputs = make_inputs(args, 0)
        cpu, cpu_outputs = cpu_baseline(args, inputs, args.cpu_images)

    import torch
    import torch.distributed as dist
    from face_detection_amd import dist as fdist
    rank, world, _ = fdist.init_from_env()
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if args.profile_only:
        args.streams = args.profile_streams
    wl = Workload(args, rank, inputs)

    if args.profile_only:
        for i in range(args.warmup + args.steps):
            wl.step(i)
        torch.cuda.synchronize()
        wl.setup_loss()
        for _ in range(10):
            wl.loss(wl.slots[0])
        torch.cuda.synchronize()
        return

    elapsed = timed_loop(torch, dist, world, wl.step, args.steps, args.warmup)
    images = args.batch * args.steps * world
    value = images / elapsed

    # the two halves on their own (same protocol), for the per-half figures of the metric
    # (auxiliary figures: at least 2000 steps each after 200 untimed ones, i.e. 30-40 ms per half, so
    # that a short --steps run reads the same steady rates: with 200 steps the fill and drain of the
    # four chains and the first milliseconds after a pause still cost 5-15 %)
    nslots = len(wl.slots)
    half_steps = max(args.steps, 2000)
    t_match = timed_loop(torch, dist, world, lambda i: wl.match(wl.slots[i % nslots]), half_steps, 200)
    t_det = timed_loop(torch, dist, world, lambda i: wl.detect(wl.slots[i % nslots]), half_steps, 200)

    # evaluation.py:67 runs the same post-processing with threshold 0: every anchor is a candidate
    thr = args.score_thr
    args.score_thr = 0.0
    n_det0 = max(args.steps // 4, 10)
    t_det0 = timed_loop(torch, dist, world, lambda i: wl.detect(wl.slots[i % nslots]), n_det0, 3)
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
        slot = wl.slots[0]
        gb = args.batch * world
        try:                                              # optional figure: never costs the line
            for _ in range(3):
                fdist.all_gather_detections(slot["dets"], slot["keep"], slot["count"], gb)
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(10):
                fdist.all_gather_detections(slot["dets"], slot["keep"], slot["count"], gb)
            torch.cuda.synchronize()
            allgather_ms = (time.perf_counter() - t0) / 10 * 1e3
            # what rank 0 received against what it computes itself from every rank's inputs
            g_dets, g_keep, g_count = fdist.all_gather_detections(slot["dets"], slot["keep"], slot["count"], gb)
            if rank == 0:
                allgather_ok = True
                mine = wl.d_ypred
                for r in range(world):
                    wl.d_ypred = torch.from_numpy(make_inputs(args, r)[4]).to(mine.device)
                    wl.detect(slot)
                    torch.cuda.synchronize()
                    sl = slice(r * args.batch, (r + 1) * args.batch)
                    allgather_ok = allgather_ok and bool(torch.equal(g_count[sl], slot["count"])) and \
                        bool(torch.equal(g_keep[sl], slot["keep"])) and \
                        bool(torch.equal(g_dets[sl].view(torch.int32), slot["dets"].view(torch.int32)))
                wl.d_ypred = mine
                wl.detect(slot)
                torch.cuda.synchronize()
        except Exception as e:
            print("rank %d: all-gather of detections failed: %r" % (rank, e), file=sys.stderr)

    # the same step on further synthetic batches (other seeds), after everything that describes
    # the reported batch: `value` stays the run above
    if rank == 0:
        stages = stage_times(wl, min(args.steps, 100))
        match_bytes, detect_bytes, n_keep = wl.algorithmic_bytes()
        gt_total = int(wl.gt_counts.sum())                # of the reported batch, before other batches load
        lat = single_batch_latency(wl)
        lat1 = single_image_latency(wl)
        pcie = pcie_times(wl)
        parity = parity_check(wl, cpu_outputs) if cpu is not None else None
    value_by_seed = [value]
    for seed in range(1, max(args.seeds, 1)):
        wl.load_inputs(make_inputs(args, rank, seed))
        t = timed_loop(torch, dist, world, wl.step, args.steps, 5)
        value_by_seed.append(images / t)

    # the stress batch (256 ground-truth boxes in every image): the match stage alone and the step
    stress = None
    if not args.stress and not args.no_stress_leg:
        wl.load_inputs(make_inputs(args, rank, 0, stress=True))
        wl.warm_all()
        t_s = timed_loop(torch, dist, world, wl.step, max(args.steps // 4, 10), 3)
        t_sm = timed_loop(torch, dist, world, lambda i: wl.match(wl.slots[i % nslots]), max(args.steps // 4, 10), 3)
        if rank == 0:
            st = stage_times(wl, 20)
            sb = 20 * int(wl.gt_counts.sum()) + 20 * wl.A * args.batch
            n_s = max(args.steps // 4, 10)
            stress = {"gt_total": int(wl.gt_counts.sum()), "match_us": st.get("match"), "force_us": st.get("force"),
                      "frac": sb / (st["match"] * 1e-6) / 1e9 / HBM_PEAK_GBS if st.get("match") else None,
                      "step_images_per_s": args.batch * n_s * world / t_s,
                      "match_encode_images_per_s": args.batch * n_s * world / t_sm}

    if rank == 0:
        KERNEL_SYMBOL = {"match": "match_geom_kernel", "force": "force_kernel", "select": "select_rows6_kernel",
                         "prune": "cand_prune_kernel", "topk": "topk_sort_kernel", "nms_mask": "nms_mask_kernel",
                         "nms_scan": "nms_scan_kernel"}
        # every kernel of a half is priced against the algorithmic bytes of that half
        half_bytes = {"match": match_bytes, "force": match_bytes, "select": detect_bytes, "prune": detect_bytes,
                      "topk": detect_bytes, "nms_mask": detect_bytes, "nms_scan": detect_bytes}
        per_kernel = {k: half_bytes[k] / (v * 1e-6) / 1e9 for k, v in stages.items() if k in half_bytes and v > 0}
        dominant = max(per_kernel, key=lambda k: stages[k])       # the longest kernel of the step
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dominant)
            except Exception:
                traffic = None
        out = {
            "metric": "images/sec (match+encode; decode+NMS) b=32 @640^2, % HBM roofline, 1-8 GPU",
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "SSD+FPN box pipeline, match+encode then decode+NMS per batch, "
                            "batch %d @%dx%d, A=%d, <=%d GT/image%s, score_thr %.2f, nms 0.45, "
                            "limit 1000" % (args.batch, args.size, args.size, wl.A, args.gmax,
                                            " (all 256)" if args.stress else "", args.score_thr),
                "batch_per_gpu": args.batch, "anchors": wl.A, "gt_total": gt_total,
                "kept_total": n_keep, "batches_in_flight": nslots, "stream_priority": args.stream_priority, "parallelism": "image-sharded x%d" % world,
            },
            "match_encode_images_per_s": args.batch * half_steps * world / t_match,
            "decode_nms_images_per_s": args.batch * half_steps * world / t_det,
            "half_steps": half_steps,
            "decode_nms_thr0_images_per_s": args.batch * n_det0 * world / t_det0,
            "kernel_us_thr0": {k: round(v, 2) for k, v in stages0.items()
                               if k in ("select", "prune", "topk", "nms_mask", "nms_scan")},
            "hbm_copy_GBs": hbm_copy_rate(torch),
            "ssd_loss_fwd_bwd_images_per_s": args.batch * n_loss * world / t_loss,
            "ssd_loss_fwd_bwd_ms": t_loss / n_loss * 1e3,
            "kernel_us": {k: round(v, 2) for k, v in stages.items()},
            "roofline": {
                "bound": "hbm", "kernel": KERNEL_SYMBOL[dominant],
                "achieved": per_kernel[dominant], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": per_kernel[dominant] / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": half_bytes[dominant],
                # the halves as they run in the timed loops (all their kernels, batches in flight):
                # algorithmic bytes of the half x its images per second / peak
                "match_encode_half_frac": match_bytes / args.batch * (args.batch * half_steps / t_match) / 1e9 / HBM_PEAK_GBS,
                "decode_nms_half_frac": detect_bytes / args.batch * (args.batch * half_steps / t_det) / 1e9 / HBM_PEAK_GBS,
            },
            "roofline_kernels": {k: {"achieved_GBs": v, "frac": v / HBM_PEAK_GBS} for k, v in per_kernel.items()},
        }
        out["value_by_seed"] = value_by_seed
        out["value_median_of_seeds"] = sorted(value_by_seed)[len(value_by_seed) // 2]
        out["single_batch_latency_ms"] = lat * 1e3
        out["single_image_decode_nms_latency_ms"] = lat1 * 1e3
        if allgather_ms is not None:
            out["allgather_detections_ms"] = allgather_ms
            out["allgather_equals_local_recompute"] = allgather_ok
        step_s = elapsed / args.steps
        pcie["images_per_s_with_transfers_serial"] = args.batch / (step_s + (pcie["h2d_ms"] + pcie["d2h_ms"]) * 1e-3)
        out["pcie"] = pcie
        if stress is not None:
            out["stress"] = stress
        parity_ok = True
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["parity"] = parity
            parity_ok = all(parity[k] == parity["images"] for k in ("match_mask_exact", "nms_kept_exact", "floats_close"))
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
