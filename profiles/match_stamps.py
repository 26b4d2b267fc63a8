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
ntile = None
# tiles per image = grid.x: find from first image's nonzero rows
stamps = raw.reshape(-1)[: (raw.size // 8) * 8].reshape(-1, 8)
valid = stamps[:, 1] > stamps[:, 0]
stamps = stamps[valid]
print("tiles with stamps:", len(stamps), " counts per image min/max:", counts.min(), counts.max())
t0 = stamps[:, 0].min()
start = (stamps[:, 0] - t0) * 0.01; end = (stamps[:, 1] - t0) * 0.01
print("kernel span (us): %.2f   tile life us: mean %.2f  p50 %.2f  p99 %.2f  max %.2f" % (end.max(), (end-start).mean(), np.median(end-start), np.percentile(end-start, 99), (end-start).max()))
print("start times us: p50 %.2f p90 %.2f max %.2f" % (np.median(start), np.percentile(start, 90), start.max()))
layer = stamps[:, 5] & 0xFF; split = stamps[:, 5] >> 8
for m in sorted(set(layer.tolist())):
    s = stamps[layer == m]
    print("layer %d (S=%d): tiles %5d  prologue cyc mean %7.0f  loop cyc mean %8.0f max %8.0f  output cyc mean %7.0f   list mean %.1f max %d  evals mean %.1f max %d  life us mean %.2f max %.2f" % (
        m, int(s[0, 5] >> 8), len(s), s[:, 2].mean(), s[:, 3].mean(), s[:, 3].max(), s[:, 4].mean(), s[:, 6].mean(), s[:, 6].max(), s[:, 7].mean(), s[:, 7].max(),
        ((s[:, 1] - s[:, 0]) * 0.01).mean(), ((s[:, 1] - s[:, 0]) * 0.01).max()))
# concurrency over time
ts = np.linspace(0, end.max(), 15)
print("tiles in flight at t:", [(round(float(x), 1), int(((start <= x) & (end > x)).sum())) for x in ts])
