#!/usr/bin/env python3
"""Timing experiment: per-tile clock stamps of the match kernel (library built with
-DSSD_EXP_STAMPS, selected through SSDBOX_LIB).  Prints per-layer phase times and the
timeline of tile start/end on the 100 MHz clock."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from face_detection_amd import _lib, synthetic as S
from face_detection_amd.runtime import Context, geometry_for
stress = "--stress" in sys.argv
lib = _lib.load()
cfg = S.reference_ssd_config(640)
rng = np.random.default_rng(1000)
boxes, labels, counts = S.make_ground_truth(rng, 32, 640, gmax=256, stress=stress)
geom = geometry_for(cfg, 0); A = geom.num_anchors
ctx = Context(0, torch.cuda.current_stream())
d = lambda x: torch.from_numpy(x).cuda()
db, dl, dc = d(boxes), d(labels), d(counts)
conf = torch.empty((32, A), device="cuda"); locs = torch.empty((32, A, 4), device="cuda")
midx = torch.zeros((32, A), dtype=torch.int32, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
for it in range(5):
    _lib.check(lib.ssdbox_match_encode(ctx.handle, geom.handle, None, A, P(db), P(dl), P(dc), 32, 256, 640.0, 0.5, P(conf), P(locs), P(midx), None))
torch.cuda.synchronize()
raw = midx.cpu().numpy().view(np.uint64)
stamps = raw.reshape(-1)[: (raw.size // 8) * 8].reshape(-1, 8)
valid = stamps[:, 1] > stamps[:, 0]
stamps = stamps[valid]
print("waves with stamps:", len(stamps), " counts per image min/max:", counts.min(), counts.max())
t0 = stamps[:, 0].min()
start = (stamps[:, 0] - t0) * 0.01; end = (stamps[:, 1] - t0) * 0.01
life = end - start
print("kernel span (us): %.2f   wave life us: mean %.2f  p50 %.2f  p99 %.2f  max %.2f" % (end.max(), life.mean(), np.median(life), np.percentile(life, 99), life.max()))
print("start times us: p50 %.2f p90 %.2f max %.2f" % (np.median(start), np.percentile(start, 90), start.max()))
layer = stamps[:, 5] & 0xFF; evals = stamps[:, 5] >> 16
cyc = stamps[:, [2, 3, 4, 6, 7]].astype(np.float64)
print("phase cycles (100 MHz memtime? see ratio): prologue / loop / far gathers+encode / store issue / store drain")
tot = cyc.sum(1)
print("  ratio memtime cycles per us of life: %.1f" % np.median(tot / np.maximum(life, 0.01)))
for m in sorted(set(layer.tolist())):
    sel = layer == m
    c = cyc[sel]
    print("layer %d (R=%d): waves %5d  prologue %7.0f  loop %7.0f (max %7.0f)  gather %7.0f  issue %6.0f  drain %7.0f   evals mean %.1f max %d  life us mean %.2f max %.2f" % (
        m, int((stamps[sel][0, 5] >> 8) & 0xFF), sel.sum(), c[:, 0].mean(), c[:, 1].mean(), c[:, 1].max(), c[:, 2].mean(), c[:, 3].mean(), c[:, 4].mean(),
        evals[sel].mean(), evals[sel].max(), life[sel].mean(), life[sel].max()))
for lo, hi in ((0, 0), (1, 1), (2, 3), (4, 8), (9, 1000)):
    sel = (evals >= lo) & (evals <= hi)
    if sel.sum():
        c = cyc[sel]
        print("evals %d-%d: waves %5d  prologue %7.0f loop %7.0f gather %7.0f issue %6.0f drain %7.0f  life %.2f us" % (lo, hi, sel.sum(), c[:, 0].mean(), c[:, 1].mean(), c[:, 2].mean(), c[:, 3].mean(), c[:, 4].mean(), life[sel].mean()))
ts = np.linspace(0, end.max(), 15)
print("waves in flight at t:", [(round(float(x), 1), int(((start <= x) & (end > x)).sum())) for x in ts])
