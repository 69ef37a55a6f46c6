"""The reference's OWN known-answer inputs pushed through the HIP path (C ABI), not only through the oracle.

tests/test_oracle_kat.py pins the oracle on these inputs on the CPU; here the same draws -- the literal seeds and shapes of the
reference's tests -- go through ``libemx.so`` and are compared with the scipy / numpy answers those tests compute.  Every test
names the reference test (file:line relative to AG-Peter/encodermap @ 2025-05-23).  Needs an MI355X.

Why these are worth a run of their own: U[0, 100)^51 rows with sigma_h = 4.5 saturate the high-d sigmoid (every pair but the
diagonal sits at 1 - O(1e-14)) while the Gram form d^2 = |xi|^2 + |xj|^2 - 2 xi.xj cancels ~1e5-sized norms -- the case where
the row-0 shift and the split-f16 Gram products of the Euclidean tiles either hold or do not.
"""
import math

import numpy as np
import pytest
import torch
from scipy.spatial.distance import cdist

pytestmark = pytest.mark.gpu
PI = math.pi
INF = float("inf")
DEFAULT_SIG = (4.5, 12, 6, 1, 2, 6)


@pytest.fixture(scope="module")
def L():
    from encodermap_amd import _lib

    _lib.require_gpu()
    _lib.load()
    return _lib


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")


def sig_metric(sig, a, b):
    def f(x, y):
        r = np.linalg.norm(x - y)
        return 1 - (1 + (2 ** (a / b) - 1) * (r / sig) ** a) ** (-b / a)
    return f


def np_sigmoid(r, sig, a, b):
    return 1 - (1 + (2 ** (a / b) - 1) * (r / sig) ** a) ** (-b / a)


def periodic_metric(periodicity):
    # tests/test_pairwise_distances.py:66-76
    def f(i, j):
        dx = np.linalg.norm(j - i)
        if dx > periodicity * 0.5:
            dx = dx - periodicity
        if dx <= -periodicity * 0.5:
            dx = dx + periodicity
        return np.abs(dx)
    return f


def cost_through_the_abi(L, x, z, periodicity, sig, split: bool, data_span=None):
    """mean over all B^2 pairs of (sig_h(d_h) - sig_l(d_l))^2 (loss_functions.py:351-367) by emx_sketchmap_tiles + emx_sketchmap_finish
    (scale 1); split: the Euclidean Gram tiles on the f16 matrix cores (emx_sketchmap_split), else the float32 VALU tiles"""
    B, Ld = z.shape
    X, Z = dev(x), dev(z)
    ws = torch.zeros(L.sketchmap_ws_bytes(B, Ld) // 4, dtype=torch.float32, device="cuda")
    dist, cen = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    dZ = torch.full((B, Ld), float("nan"), device="cuda")
    sp = None
    if split:
        sp = torch.zeros(L.sketchmap_split_bytes(B, x.shape[1]), dtype=torch.uint8, device="cuda")
        L.sketchmap_split(X, sp)
    L.sketchmap_tiles(X, Z, periodicity, sig, 0, B, ws, sp, data_span=data_span)
    L.sketchmap_finish(B, x.shape[1], Z, sig, 1.0, 0.0, 0, B, dist, cen, dZ, ws, periodicity=periodicity)
    torch.cuda.synchronize()
    g = dZ.cpu().numpy()
    assert np.isfinite(g).all()
    return float(dist), g


@pytest.mark.parametrize("split", [False, True])
def test_sketchmap_cost_non_periodic_reference_draws(L, split):
    """tests/test_losses.py:196-229: RandomState(0), highd U[0, 100)^{256 x 51}, lowd U[0, 10)^{256 x 2}, default sigmoid
    parameters; expected = mean((cdist_sig_h - cdist_sig_l)^2) by scipy.  The reference asserts TensorFlow's default tolerance
    (1e-6, float32 run); here 1e-5 relative (the GPU tests' cost tolerance), and the latent gradient against the float64 oracle."""
    from oracle import direct_ops as D

    rng = np.random.RandomState(0)
    highd = rng.random_sample((256, 51)).astype("float32") * 100
    lowd = rng.random_sample((256, 2)).astype("float32") * 10
    sig_h = cdist(highd, highd, sig_metric(*DEFAULT_SIG[:3]))
    sig_l = cdist(lowd, lowd, sig_metric(*DEFAULT_SIG[3:]))
    cost_scipy = np.mean(np.square(sig_h - sig_l))
    cost, g = cost_through_the_abi(L, highd, lowd, INF, DEFAULT_SIG, split)
    assert cost == pytest.approx(cost_scipy, rel=1e-5)
    _, g_o = D.sketchmap_fwd_bwd(highd, lowd, INF, DEFAULT_SIG, scale=1.0)
    np.testing.assert_allclose(g, g_o, rtol=0, atol=2e-5 * np.abs(g_o).max() + 1e-12)


@pytest.mark.parametrize("phases", [False, True])
def test_sketchmap_cost_periodic_reference_draws(L, phases):
    """tests/test_losses.py:231-280: RandomState(1), highd U[-pi, pi)^{256 x 51}, lowd U[0, 10)^{256 x 2}, periodicity 2 pi; expected:
    the test's numpy restatement (|b - a|, min(d, 2 pi - d), + 1e-16 where 0, norm over axis 2) and the wrapped scipy metric
    (the reference holds the two to atol 1e-3 of each other).  phases: the tile pass on 32-bit phases (the span of the data is
    known to be below 1.5 periods), else the float form of min(|d|, p - |d|)."""
    rng = np.random.RandomState(1)
    highd = (rng.random_sample((256, 51)).astype("float32") * 2 * np.pi) - np.pi
    lowd = rng.random_sample((256, 2)).astype("float32") * 10
    a = np.expand_dims(highd.astype(np.float64), axis=1)
    b = np.expand_dims(highd.astype(np.float64), axis=0)
    d = np.abs(b - a)
    d = np.minimum(d, 2 * np.pi - d)
    d = d + np.equal(d, 0.0) * 1e-16
    d = np.linalg.norm(d, axis=2)
    sig_h = np_sigmoid(d, *DEFAULT_SIG[:3])
    sig_l = cdist(lowd, lowd, sig_metric(*DEFAULT_SIG[3:]))
    cost_numpy = np.mean(np.square(sig_h - sig_l))
    span = float(highd.max() - highd.min()) if phases else None
    cost, _ = cost_through_the_abi(L, highd, lowd, 2 * np.pi, DEFAULT_SIG, False, data_span=span)
    assert cost == pytest.approx(cost_numpy, rel=1e-5)

    def wrapped_sig(x, y):  # periodicity_closure_w_sigmoid
        dv = np.abs(x - y)
        dv = np.minimum(dv, 2 * np.pi - dv)
        return np_sigmoid(np.linalg.norm(dv), *DEFAULT_SIG[:3])
    cost_scipy = np.mean(np.square(cdist(highd, highd, wrapped_sig) - sig_l))
    assert abs(cost - cost_scipy) < 1e-3


def test_sigmoid_of_pairwise_dist_reference_draws(L):
    """tests/test_pairwise_distances.py:81-93: np.random.seed(1); random((100, 2)) * 360; sigmoid(4.5, 12, 6) of pairwise_dist vs
    scipy at atol 1e-3.  The distances come from emx_pairwise_dist (the materialising a2 variant); the sigmoid of the test is
    applied to them in float64 on the host, so the comparison prices the kernel's distances alone."""
    np.random.seed(1)
    points = np.random.random((100, 2)).astype(np.float32) * 360.0
    expect = cdist(points, points, sig_metric(*DEFAULT_SIG[:3]))
    out = torch.empty(100, 100, device="cuda")
    L.pairwise_dist(dev(points), INF, out)
    d = out.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(d, cdist(points.astype(np.float64), points.astype(np.float64)), rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(np_sigmoid(d, *DEFAULT_SIG[:3]), expect, atol=1e-3)


def test_periodic_pairwise_reference_draws(L):
    """tests/test_pairwise_distances.py:110-135: [20, 40, 340] with p = 360 and seed(1) random((100, 1)) * 360 against the wrapped
    scipy metric, atol 1e-5 -- plus the literal answers [[0, 20, 40], [20, 0, 60], [40, 60, 0]]."""
    m = periodic_metric(360.0)
    pts = np.vstack([20.0, 40.0, 340.0]).astype(np.float32)
    out = torch.empty(3, 3, device="cuda")
    L.pairwise_dist(dev(pts), 360.0, out)
    np.testing.assert_allclose(out.cpu().numpy(), cdist(pts, pts, m), atol=1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), [[0, 20, 40], [20, 0, 60], [40, 60, 0]], atol=1e-5)
    np.random.seed(1)
    pts = np.random.random((100, 1)).astype(np.float32) * 360.0
    out = torch.empty(100, 100, device="cuda")
    L.pairwise_dist(dev(pts), 360.0, out)
    # (float32 kernel on values up to 360: half an ulp of 360 is 1.5e-5; the reference's own float32 run sits at the same place)
    np.testing.assert_allclose(out.cpu().numpy(), cdist(pts, pts, m), rtol=2e-6, atol=4e-5)


def test_periodic_distance_literal_through_the_cost_kernel(L):
    """tests/test_pairwise_distances.py:98-108: periodic_distance([0, 0, 0], [pi/2, pi, 3 pi/2], 2 pi) == [pi/2, pi, pi/2] -- through
    emx_periodic_cost_fwd_bwd (mean_abs of one row of three columns = the mean of those three numbers = 2 pi / 3), and the doctest
    value of misc/distances.py:130-137: periodic_distance(1.5, -3.1) = 1.68318531."""
    a = dev([[0.0, 0.0, 0.0]])
    b = dev([[PI / 2, PI, 3 / 2 * PI]])
    loss = torch.zeros(1, device="cuda")
    ws = torch.zeros(4096, device="cuda")
    L.periodic_cost_fwd_bwd(b, 0, 0, False, a, 3, 2 * PI, "mean_abs", 1.0, 0, 1, loss, None, ws)
    assert float(loss) == pytest.approx(2 * PI / 3, rel=2e-6)
    L.periodic_cost_fwd_bwd(dev([[1.5]]), 0, 0, False, dev([[-3.1]]), 1, 2 * PI, "mean_abs", 1.0, 0, 1, loss, None, ws)
    assert float(loss) == pytest.approx(1.68318531, abs=2e-6)
