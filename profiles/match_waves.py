#!/usr/bin/env python3
"""Timing experiment: when every wave of one match_geom_kernel launch starts, ends its prologue and
ends (library built with -DSSD_EXP_WGTRACE, selected through SSDBOX_LIB):
  make -C face_detection_amd/csrc BUILD=build_wg OUT=$PWD/face_detection_amd/libssdbox_wg.so EXTRA=-DSSD_EXP_WGTRACE
  SSDBOX_LIB=$PWD/face_detection_amd/libssdbox_wg.so python3 profiles/match_waves.py [--stress]"""
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
trace = torch.zeros((32 * 4096 * 4 * 4,), dtype=torch.int64, device="cuda")
P = lambda t: ctypes.c_void_p(t.data_ptr())
lib.ssdbox_debug_set_wgtrace.argtypes = [ctypes.c_void_p]
call = lambda: _lib.check(lib.ssdbox_match_encode(ctx.handle, geom.handle, None, A, P(db), P(dl), P(dc), 32, 256, 640.0, 0.5, P(conf), P(locs), None, None))
for it in range(5):
    call()
torch.cuda.synchronize()
lib.ssdbox_debug_set_wgtrace(P(trace))
call()
torch.cuda.synchronize()
lib.ssdbox_debug_set_wgtrace(None)
st = trace.cpu().numpy().view(np.uint64).reshape(-1, 4)
st = st[st[:, 2] > 0]
t0 = st[:, 0].min()
start, pro, end = [(st[:, i] - t0) * 0.01 for i in range(3)]
R = (st[:, 3] >> 8) & 0xFF; nbox = st[:, 3] & 0xFF; nlist = st[:, 3] >> 16
print("waves %d  span %.2f us" % (len(st), end.max()))
print("start  p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(start, [10, 50, 90, 99, 100])))
print("end    p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(end, [10, 50, 90, 99, 100])))
print("life   mean %.2f p50 %.2f p90 %.2f max %.2f   prologue mean %.2f p90 %.2f" % (
    (end - start).mean(), np.median(end - start), np.percentile(end - start, 90), (end - start).max(),
    (pro - start).mean(), np.percentile(pro - start, 90)))
for r in sorted(set(R.tolist())):
    m = R == r
    print("  R=%d waves %5d  life mean %.2f max %.2f  prologue %.2f  listed boxes mean %.1f max %d  last end %.2f  first start %.2f last start %.2f" % (
        r, m.sum(), (end - start)[m].mean(), (end - start)[m].max(), (pro - start)[m].mean(), nlist[m].mean(), nlist[m].max(),
        end[m].max(), start[m].min(), start[m].max()))
# how many waves are alive / past their prologue over time
ts = np.arange(0, end.max() + 0.5, 1.0)
print("t(us): alive, storing(past prologue)")
for t in ts:
    print("  %5.1f  %5d  %5d" % (t, ((start <= t) & (end > t)).sum(), ((pro <= t) & (end > t)).sum()))
