import sys, time, ctypes
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from face_detection_amd import synthetic as S, _lib
from face_detection_amd.utils import anchor_utils, box_utils
from face_detection_amd.runtime import Session, DevArray, _ptr
cfg = S.reference_ssd_config(640, "config")
rng = np.random.default_rng(5)
boxes, labels, counts = S.make_ground_truth(rng, 1, 640, gmax=64)
y = S.make_head_outputs(rng, cfg, boxes, counts)[0]
anchors = anchor_utils.generate_anchors(cfg)
dec = box_utils.decode(anchors, np.ascontiguousarray(y[:, 2:]))
m = y[:, 1] > 0.05
cb, cs = dec[m], y[m, 1]
def T(label, fn, reps=50):
    for _ in range(5): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    print("%-28s %.4f ms" % (label, (time.perf_counter() - t0) / reps * 1e3))
T("compute_nms numpy", lambda: box_utils.compute_nms(cb, cs, 0.45, 1000))
sess = Session(cb)
ctx = sess.ctx
T("put boxes+scores", lambda: (sess.put(cb, np.float32), sess.put(cs, np.float32)))
b, _ = sess.put(cb, np.float32); s, _ = sess.put(cs, np.float32)
keep = sess.empty((1000,), np.int32); count = sess.empty((1,), np.int32)
def kern():
    _lib.check(ctx.lib.ssdbox_compute_nms(ctx.handle, _ptr(b), _ptr(s), len(cb), 0.45, 1000, _ptr(keep), _ptr(count)))
T("kernel enqueue", kern)
def kern_sync():
    kern(); ctx.synchronize()
T("kernel + sync", kern_sync)
T("host(count)", lambda: sess.host(count))
T("host(keep)", lambda: sess.host(keep))
T("empty+free", lambda: sess.empty((1000,), np.int32))
