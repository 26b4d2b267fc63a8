"""SSD loss with hard-negative mining (reference utils/losses.py:97-181).  The reference has no
tests or fixtures for it and TensorFlow is not installed: parity unpinned.  The numpy oracle
follows the reference line by line and is pinned by hand-derived values; the C oracle uses the
kernels' logarithm, so the mined-negative mask of the HIP path must equal it exactly, sums and
gradients to 1e-6 relative, and the numpy oracle to 1e-5 (the tolerance stated for floats)."""
import numpy as np
import pytest

from oracle import box_oracle as O
from oracle import c_oracle as C

F32 = np.float32


def make_batch(rng, B, A, pos_rate=0.01, ties=False):
    pos = (rng.uniform(size=(B, A)) < pos_rate).astype(F32)
    yt = np.zeros((B, A, 6), F32)
    yt[..., 1] = pos
    yt[..., 0] = 1 - pos
    yt[..., 2:] = (rng.normal(0, 1, (B, A, 4)) * pos[..., None]).astype(F32)
    logit = rng.normal(-3, 1.5, (B, A))
    if ties:
        logit = np.round(logit * 2) / 2            # few distinct values: many equal losses
    p1 = (1 / (1 + np.exp(-logit))).astype(F32)
    yp = np.zeros((B, A, 6), F32)
    yp[..., 1] = p1
    yp[..., 0] = F32(1) - p1
    yp[..., 2:] = rng.normal(0, 1.2, (B, A, 4)).astype(F32)
    return yt, yp


def test_loss_hand_values():
    # anchor 0 positive, 1..3 negative; k = min(3 * 1, 3 nonzero) = 3: all negatives kept
    yt = np.array([[[0, 1, .1, .2, .3, .4], [1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0]]], F32)
    yp = np.array([[[.2, .8, .1, .2, .3, 2.4], [.9, .1, 0, 0, 0, 0], [.5, .5, 0, 0, 0, 0], [.99, .01, 0, 0, 0, 0]]], F32)
    total, keep, cls, npos = O.ssd_loss(yt, yp, details=True)
    want = -np.log(.8) - np.log(.9) - np.log(.5) - np.log(.99) + (2.0 - 0.5)
    assert npos == 1 and keep.tolist() == [[False, True, True, True]]
    np.testing.assert_allclose(total, [want], rtol=1e-6)
    # neg_pos_ratio 1: only the hardest negative (p_bg = .5) stays
    total, keep, _, _ = O.ssd_loss(yt, yp, neg_pos_ratio=1, details=True)
    assert keep.tolist() == [[False, False, True, False]]
    np.testing.assert_allclose(total, [-np.log(.8) - np.log(.5) + 1.5], rtol=1e-6)
    # no positives: k = n_neg_min, the division is by max(1, 0)
    yt0 = yt.copy()
    yt0[0, 0] = [1, 0, 0, 0, 0, 0]
    total, keep, _, npos = O.ssd_loss(yt0, yp, n_neg_min=2, details=True)
    assert npos == 0 and keep.tolist() == [[True, False, True, False]]      # -log(.2) and -log(.5)
    np.testing.assert_allclose(total, [-np.log(.2) - np.log(.5)], rtol=1e-6)
    total = O.ssd_loss(yt0, yp)
    assert total.tolist() == [0.0]
    # batch factor: two identical images, each loss is (sum of both) / n_pos * 2 ... = same value
    total2 = O.ssd_loss(np.concatenate([yt, yt]), np.concatenate([yp, yp]))
    np.testing.assert_allclose(total2, [want, want], rtol=1e-6)
    # equal losses: the lower index wins (tf.nn.top_k order)
    yp_t = yp.copy()
    yp_t[0, 1:, 0] = .5
    yp_t[0, 1:, 1] = .5
    _, keep, _, _ = O.ssd_loss(yt, yp_t, neg_pos_ratio=2, details=True)
    assert keep.tolist() == [[False, True, True, False]]
    # probabilities of zero are clipped at 1e-15 (:78)
    yp_z = yp.copy()
    yp_z[0, 0, 1] = 0.0
    _, _, cls, _ = O.ssd_loss(yt, yp_z, details=True)
    np.testing.assert_allclose(cls[0, 0], -np.log(F32(1e-15)), rtol=1e-6)


def test_loss_gradient_matches_finite_differences():
    rng = np.random.default_rng(4)
    yt, yp = make_batch(rng, 2, 40, pos_rate=0.15)
    total, keep, _, npos = O.ssd_loss(yt, yp, details=True)
    g = O.ssd_loss_grad(yt, yp, keep, npos)
    yt64, yp64 = yt.astype(np.float64), yp.astype(np.float64)

    def mean_loss(ypv):
        # float64 restatement with the mask frozen
        cls = -(yt64[..., :2] * np.log(np.maximum(ypv[..., :2], 1e-15))).sum(-1)
        d = yt64[..., 2:] - ypv[..., 2:]
        loc = np.where(np.abs(d) < 1, 0.5 * d * d, np.abs(d) - 0.5).sum(-1)
        pos = yt64[..., 1]
        tot = ((cls * pos).sum(-1) + (cls * keep).sum(-1) + (loc * pos).sum(-1)) / max(1.0, npos) * yt.shape[0]
        return tot.mean()

    eps = 1e-6
    for (b, a, c) in [(0, 3, 0), (0, 3, 1), (1, 7, 2), (1, 20, 5), (0, int(np.argmax(yt[0, :, 1])), 3)]:
        hi, lo = yp64.copy(), yp64.copy()
        hi[b, a, c] += eps
        lo[b, a, c] -= eps
        fd = (mean_loss(hi) - mean_loss(lo)) / (2 * eps)
        assert abs(fd - g[b, a, c]) <= 1e-4 * max(1.0, abs(fd))


@pytest.mark.parametrize("seed,B,A,kw", [(0, 3, 500, {}), (1, 2, 3000, {"neg_pos_ratio": 1}),
                                         (2, 4, 700, {"n_neg_min": 50, "alpha": 0.5})])
def test_c_oracle_matches_numpy(seed, B, A, kw):
    rng = np.random.default_rng(seed)
    yt, yp = make_batch(rng, B, A, ties=(seed == 2))
    total, keep, cls, npos = O.ssd_loss(yt, yp, details=True, **kw)
    c_total, c_keep, c_cls, c_npos, c_grad = C.ssd_loss(yt, yp, grad=True, **kw)
    assert npos == c_npos
    np.testing.assert_allclose(c_cls, cls, rtol=1e-6, atol=1e-7)
    if seed != 2:        # with ties a 1-ULP difference of the logarithms may reorder equal-looking values
        np.testing.assert_array_equal(c_keep, keep)
    assert c_keep.sum() == keep.sum()
    np.testing.assert_allclose(c_total, total, rtol=1e-5)
    g = O.ssd_loss_grad(yt, yp, c_keep, c_npos, kw.get("alpha", 1.0))
    np.testing.assert_allclose(c_grad, g, rtol=1e-6, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,B,A,kw,ties", [
    (0, 3, 500, {}, False), (1, 2, 3000, {"neg_pos_ratio": 1}, False),
    (2, 4, 777, {"n_neg_min": 50, "alpha": 0.5}, True), (3, 1, 64, {}, True),
    (4, 5, 2049, {"neg_pos_ratio": 7}, True), (5, 2, 300, {"neg_pos_ratio": 0}, False),
    (6, 3, 6002, {"neg_pos_ratio": 2}, True)])
def test_hip_loss_matches_c_oracle(seed, B, A, kw, ties):
    import face_detection_amd as fda
    rng = np.random.default_rng(seed)
    yt, yp = make_batch(rng, B, A, ties=ties)
    total, keep, grad = fda.ssd_loss_batch(yt, yp, return_keep=True, return_grad=True, **kw)
    c_total, c_keep, _, c_npos, c_grad = C.ssd_loss(yt, yp, grad=True, **kw)
    np.testing.assert_array_equal(keep, c_keep)
    np.testing.assert_allclose(total, c_total, rtol=1e-6)
    np.testing.assert_array_equal(grad.view(np.uint32), c_grad.view(np.uint32))
    np.testing.assert_allclose(total, O.ssd_loss(yt, yp, **kw), rtol=1e-5)
    only = fda.ssd_loss_batch(yt, yp, **kw)
    np.testing.assert_array_equal(only, total)


@pytest.mark.gpu
def test_hip_loss_edge_cases():
    import face_detection_amd as fda
    from face_detection_amd.utils.losses import SSDLoss
    rng = np.random.default_rng(9)
    yt, yp = make_batch(rng, 2, 400)
    # no positive anywhere: nothing is mined with n_neg_min = 0, loss 0
    yt0 = yt.copy()
    yt0[..., 1] = 0
    yt0[..., 0] = 1
    total, keep = fda.ssd_loss_batch(yt0, yp, return_keep=True)
    assert not keep.any() and total.tolist() == [0.0, 0.0]
    # ... and n_neg_min of them otherwise
    total, keep = fda.ssd_loss_batch(yt0, yp, n_neg_min=37, return_keep=True)
    c_total, c_keep, _, _ = C.ssd_loss(yt0, yp, n_neg_min=37)
    assert keep.sum() == 37
    np.testing.assert_array_equal(keep, c_keep)
    np.testing.assert_allclose(total, c_total, rtol=1e-6)
    # every loss equal: the first k in index order are mined
    yp_e = yp.copy()
    yp_e[..., 0] = 0.75
    yp_e[..., 1] = 0.25
    total, keep = fda.ssd_loss_batch(yt, yp_e, return_keep=True)
    c_total, c_keep, _, _ = C.ssd_loss(yt, yp_e)
    np.testing.assert_array_equal(keep, c_keep)
    np.testing.assert_allclose(total, c_total, rtol=1e-6)
    # ... across many images and several chunks per image (tie order = flat index order)
    yt_m, yp_m = make_batch(rng, 37, 4100, pos_rate=0.02)
    yp_m[..., 0] = 0.75
    yp_m[..., 1] = 0.25
    total, keep = fda.ssd_loss_batch(yt_m, yp_m, neg_pos_ratio=5, return_keep=True)
    c_total, c_keep, _, _ = C.ssd_loss(yt_m, yp_m, neg_pos_ratio=5)
    np.testing.assert_array_equal(keep, c_keep)
    np.testing.assert_allclose(total, c_total, rtol=1e-6)
    flat = keep.reshape(-1)
    assert flat.any() and not flat[np.flatnonzero(flat)[-1] + 1:].any()
    # zero probabilities are clipped, the gradient there is zero
    yp_z = yp.copy()
    yp_z[0, :50, 0] = 0.0
    total, grad = SSDLoss().call_with_grad(yt, yp_z)
    c_total, _, _, _, c_grad = C.ssd_loss(yt, yp_z, grad=True)
    np.testing.assert_allclose(total, c_total, rtol=1e-6)
    np.testing.assert_array_equal(grad.view(np.uint32), c_grad.view(np.uint32))
    with pytest.raises(ValueError):
        fda.ssd_loss_batch(yt, yp[:, :10])


@pytest.mark.gpu
def test_hip_loss_full_size_batch32():
    """b=32, A=102000 (the bench shape): mask exact against the C oracle, sums to 1e-6."""
    import face_detection_amd as fda
    rng = np.random.default_rng(21)
    yt, yp = make_batch(rng, 32, 102000, pos_rate=0.002)
    total, keep = fda.ssd_loss_batch(yt, yp, return_keep=True)
    c_total, c_keep, _, c_npos = C.ssd_loss(yt, yp)
    assert keep.sum() == c_keep.sum() == 3 * int(c_npos)
    np.testing.assert_array_equal(keep, c_keep)
    np.testing.assert_allclose(total, c_total, rtol=1e-6)
