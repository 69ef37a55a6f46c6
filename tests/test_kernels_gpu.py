"""Parity of every HIP kernel (through the C ABI) against the float64 oracle.  Needs an MI355X.

Tolerances (float32 kernels vs float64 oracle): loss 1e-5 relative (north star: 1e-4), gradients
2e-5 relative to the largest gradient entry, element-wise transforms 2e-6 absolute on O(1) values.
"""
import math

import numpy as np
import pytest
import torch

from oracle import direct_ops as D

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


def dev(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda")


def make_xz(rng, B, Dd, Ld, periodicity, spread=1.0):
    if periodicity == INF:
        x = rng.normal(0, spread, (B, Dd))
    else:
        x = rng.uniform(-periodicity / 2, periodicity / 2, (B, Dd))
    z = rng.normal(0, 1.0, (B, Ld))
    return x.astype(np.float32), z.astype(np.float32)


def run_sketchmap(L, x, z, periodicity, sig, scale, center_scale, rb, re, data_span=None):
    B, Ld = z.shape
    X, Z = dev(x), dev(z)
    ws = torch.zeros(L.sketchmap_ws_bytes(B, Ld) // 4, dtype=torch.float32, device="cuda")
    dist = torch.zeros(1, device="cuda")
    cen = torch.zeros(1, device="cuda") if center_scale is not None else None
    dZ = torch.full((B, Ld), float("nan"), device="cuda")
    L.sketchmap_fwd_bwd(X, Z, periodicity, sig, scale, center_scale, rb, re, dist, cen, dZ, ws, data_span=data_span)
    torch.cuda.synchronize()
    return float(dist), (float(cen) if cen is not None else None), dZ.cpu().numpy().astype(np.float64)


SK_CASES = [
    # B, D, L, periodicity, sig
    (2, 1, 2, INF, DEFAULT_SIG),
    (3, 2, 2, 2 * PI, DEFAULT_SIG),
    (64, 10, 2, 2 * PI, DEFAULT_SIG),
    (256, 51, 2, INF, DEFAULT_SIG),
    (256, 51, 2, 2 * PI, DEFAULT_SIG),
    (257, 33, 2, INF, (6.0, 12, 6, 1, 2, 6)),
    (1000, 128, 2, INF, (16.0, 12, 6, 1, 2, 6)),
    (1024, 128, 2, 2 * PI, (8.0, 12, 6, 1, 2, 6)),
    (300, 7, 3, INF, (2.0, 3, 6, 1, 2, 6)),       # general-exponent path, odd D, L=3
    (200, 12, 1, 360.0, (90.0, 10, 5, 1.5, 4, 2)),  # general path, L=1, period 360
    (130, 4, 4, INF, (1.5, 12, 6, 1, 2, 6)),      # L=4 runtime-L instantiation
    (500, 20, 2, INF, (3.0, 2, 2, 1, 1.5, 3)),    # a_l < 2: singular S'/d needs the d==0 mask
    # latent widths 5 .. 8 (the reference stacks any n_neurons, models.py:3133-3160): the eight-column instantiations of the
    # 32 x 32 VALU tiles (with split rows too: see test_sketchmap_on_split_rows)
    (300, 16, 5, INF, (4.0, 12, 6, 1, 2, 6)),
    (700, 33, 8, 2 * PI, DEFAULT_SIG),
    (260, 10, 7, INF, (2.0, 3, 6, 1, 2, 6)),
    # any latent width (round 3): beyond eight columns the tiles walk the latent eight columns at a time (SK_LT_ANY)
    (300, 16, 9, INF, (4.0, 12, 6, 1, 2, 6)),
    (700, 33, 16, 2 * PI, DEFAULT_SIG),
    (333, 10, 21, INF, (2.0, 3, 6, 1, 2, 6)),
    (150, 40, 64, 2 * PI, (8.0, 12, 6, 2.5, 2, 6)),
]


@pytest.mark.parametrize("B,Dd,Ld,periodicity,sig", SK_CASES)
def test_sketchmap_full_batch(L, B, Dd, Ld, periodicity, sig):
    rng = np.random.default_rng(B * 7 + Dd)
    spread = sig[0] / math.sqrt(2 * Dd) if periodicity == INF else 1.0
    x, z = make_xz(rng, B, Dd, Ld, periodicity, spread)
    if B > 8:
        z[5] = z[7]      # coincident latent points
        x[2] = x[3]      # coincident high-d points
    loss_o, g_o = D.sketchmap_fwd_bwd(x, z, periodicity, sig, scale=500.0)
    cen_o = 1e-4 * np.mean(z.astype(np.float64) ** 2)
    g_o = g_o + 2e-4 * z.astype(np.float64) / z.size
    loss, cen, g = run_sketchmap(L, x, z, periodicity, sig, 500.0, 1e-4, 0, B)
    assert loss == pytest.approx(loss_o, rel=1e-5, abs=1e-9)
    assert cen == pytest.approx(cen_o, rel=1e-5)
    np.testing.assert_allclose(g, g_o, rtol=0, atol=2e-5 * np.abs(g_o).max() + 1e-12)


@pytest.mark.parametrize("B,Dd,Ld,periodicity,sig,shift", [
    (64, 10, 2, 2 * PI, DEFAULT_SIG, 0.0),                      # 32 x 32 tiles
    (300, 12, 2, 2 * PI, DEFAULT_SIG, PI),                      # angles in [0, 2 pi)
    (1024, 128, 2, 2 * PI, (8.0, 12, 6, 1, 2, 6), 0.0),         # 64 x 64 tiles
    (700, 33, 8, 360.0, (150.0, 12, 6, 1, 2, 6), 180.0),        # degrees in [0, 360), wide latent, D not a multiple of 4
    (200, 12, 1, 360.0, (90.0, 10, 5, 1.5, 4, 2), 0.0),         # general sigmoid exponents
    (600, 7, 3, 2 * PI, DEFAULT_SIG, 0.25 * PI),                # span 1.25 periods: data in [-pi + pi/4, pi + pi/2)
])
def test_sketchmap_periodic_fixed_point_tiles(L, B, Dd, Ld, periodicity, sig, shift):
    """Periodic data whose span is known to be below 1.5 periods (``data_span``) run the tile pass on 32-bit phases (wrapping
    integer differences); without it the float form of the reference's min(|d|, p - |d|).  Both are the float64 oracle to the
    kernel's tolerance, and they are not the same launch (the results differ in the last bits)."""
    rng = np.random.default_rng(B + Dd)
    x, z = make_xz(rng, B, Dd, Ld, periodicity, 1.0)
    x = x + np.float32(shift)
    if shift == 0.25 * PI:
        x[::2] += np.float32(0.25 * PI)
    x[0, 0], x[1, 0] = x.min(), x.min() + np.float32(periodicity / 2)        # a pair exactly half a period apart
    x[2] = x[3]
    span = float(x.max() - x.min())
    assert span < 1.5 * periodicity
    loss_o, g_o = D.sketchmap_fwd_bwd(x, z, periodicity, sig, scale=500.0)
    res = {}
    for name, ds in (("float", None), ("phase", span)):
        loss, _, g = run_sketchmap(L, x, z, periodicity, sig, 500.0, None, 0, B, data_span=ds)
        assert loss == pytest.approx(loss_o, rel=1e-5, abs=1e-9), name
        np.testing.assert_allclose(g, g_o, rtol=0, atol=2e-5 * np.abs(g_o).max() + 1e-12, err_msg=name)
        res[name] = (loss, g)
    assert res["float"][0] != res["phase"][0] or not np.array_equal(res["float"][1], res["phase"][1])
    # row blocks (the asymmetric rectangle path) on phases
    rb, re = B // 3, B - 5
    lo_o, gb_o = D.sketchmap_fwd_bwd(x, z, periodicity, sig, 500.0, rb, re)
    lo, _, gb = run_sketchmap(L, x, z, periodicity, sig, 500.0, None, rb, re, data_span=span)
    assert lo == pytest.approx(lo_o, rel=1e-5)
    np.testing.assert_allclose(gb[rb:re], gb_o[rb:re], rtol=0, atol=2e-5 * np.abs(gb_o).max())


def test_sketchmap_wide_span_keeps_the_reference_formula(L):
    """Data that span more than 1.5 periods: the reference's min(|d|, p - |d|) goes NEGATIVE beyond one period and is squared
    -- not the wrapped distance.  The float tiles follow it for any span; a span of 1.5 periods or more (or none given) selects
    them.  The phase tiles would compute the wrapped distance: forced on this data (by understating the span) they must
    disagree with the reference formula -- the gate is what keeps parity."""
    rng = np.random.default_rng(9)
    B, Dd, p = 300, 6, 2 * PI
    x = rng.uniform(-1.6 * p, 1.6 * p, (B, Dd)).astype(np.float32)
    z = rng.normal(0, 1, (B, 2)).astype(np.float32)
    span = float(x.max() - x.min())
    assert span > 1.5 * p
    loss_o, g_o = D.sketchmap_fwd_bwd(x, z, p, DEFAULT_SIG, scale=500.0)
    for ds in (None, span, 1.5 * p):
        loss, _, g = run_sketchmap(L, x, z, p, DEFAULT_SIG, 500.0, None, 0, B, data_span=ds)
        assert loss == pytest.approx(loss_o, rel=1e-5)
        np.testing.assert_allclose(g, g_o, rtol=0, atol=2e-5 * np.abs(g_o).max())
    loss_w, _, _ = run_sketchmap(L, x, z, p, DEFAULT_SIG, 500.0, None, 0, B, data_span=1.0)
    assert abs(loss_w - loss_o) > 1e-3 * abs(loss_o)


@pytest.mark.parametrize("B,Dd,periodicity,blocks", [
    (512, 32, INF, [(0, 128), (128, 512)]),
    (512, 32, 2 * PI, [(0, 100), (100, 333), (333, 512)]),
    (1030, 16, INF, [(0, 515), (515, 1030)]),
])
def test_sketchmap_row_blocks_sum_to_full(L, B, Dd, periodicity, blocks):
    rng = np.random.default_rng(3)
    x, z = make_xz(rng, B, Dd, 2, periodicity, 1.0)
    sig = (2.0 * math.sqrt(Dd) if periodicity == INF else 4.5, 12, 6, 1, 2, 6)
    loss_full, _, g_full = run_sketchmap(L, x, z, periodicity, sig, 500.0, 1e-4, 0, B)
    loss_sum, cen_sum = 0.0, 0.0
    g_sum = np.zeros_like(g_full)
    for rb, re in blocks:
        lo_o, g_o = D.sketchmap_fwd_bwd(x, z, periodicity, sig, 500.0, rb, re)
        lo, ce, g = run_sketchmap(L, x, z, periodicity, sig, 500.0, 1e-4, rb, re)
        assert lo == pytest.approx(lo_o, rel=1e-5)
        assert np.all(np.isnan(g[:rb])) and np.all(np.isnan(g[re:]))   # rows outside the block untouched
        g_o = g_o[rb:re] + 2e-4 * z[rb:re].astype(np.float64) / z.size
        np.testing.assert_allclose(g[rb:re], g_o, rtol=0, atol=2e-5 * np.abs(g_o).max())
        loss_sum += lo
        cen_sum += ce
        g_sum[rb:re] = g[rb:re]
    assert loss_sum == pytest.approx(loss_full, rel=2e-6)
    np.testing.assert_allclose(g_sum, g_full, rtol=0, atol=1e-5 * np.abs(g_full).max())


def test_sketchmap_deterministic_and_rejects(L):
    rng = np.random.default_rng(0)
    x, z = make_xz(rng, 777, 24, 2, INF, 1.0)
    a = run_sketchmap(L, x, z, INF, DEFAULT_SIG, 500.0, 1e-4, 0, 777)
    b = run_sketchmap(L, x, z, INF, DEFAULT_SIG, 500.0, 1e-4, 0, 777)
    assert a[0] == b[0] and np.array_equal(a[2], b[2])           # bitwise reproducible
    X = dev(x)
    with pytest.raises(L.EmxError, match="workspace"):              # (nine latent columns are fine since round 3 -- with room for them)
        L.sketchmap_fwd_bwd(X, dev(np.zeros((777, 9), np.float32)), INF, DEFAULT_SIG, 1.0, None, 0, 777, torch.zeros(1, device="cuda"), None,
                            torch.zeros(777, 9, device="cuda"), torch.zeros(L.sketchmap_ws_bytes(777, 2) // 4, device="cuda"))
    with pytest.raises(L.EmxError, match="workspace"):
        L.sketchmap_fwd_bwd(X, dev(z), INF, DEFAULT_SIG, 1.0, None, 0, 777, torch.zeros(1, device="cuda"), None,
                            torch.zeros(777, 2, device="cuda"), torch.empty(16, device="cuda"))


def test_sketchmap_full_size_properties(L):
    """BASELINE config C2 size (B=4096, D=128): size-independent properties instead of an O(B^2 D) oracle run:
    (i) row-block partials add up to the full-batch value, (ii) translation invariance of X and Z,
    (iii) the gradient sums to zero over the batch (Newton's third law of the pair terms),
    (iv) a 512-row sub-batch embedded twice gives the analytically related loss."""
    rng = np.random.default_rng(1)
    B, Dd = 4096, 128
    x, z = make_xz(rng, B, Dd, 2, INF, 1.0)
    sig = (16.0, 12, 6, 1, 2, 6)
    loss, _, g = run_sketchmap(L, x, z, INF, sig, 500.0, None, 0, B)
    parts = [run_sketchmap(L, x, z, INF, sig, 500.0, None, rb, rb + 1024) for rb in range(0, B, 1024)]
    assert sum(p[0] for p in parts) == pytest.approx(loss, rel=2e-6)
    g_parts = np.concatenate([p[2][i * 1024:(i + 1) * 1024] for i, p in enumerate(parts)])
    np.testing.assert_allclose(g_parts, g, rtol=0, atol=1e-5 * np.abs(g).max())
    loss_t, _, g_t = run_sketchmap(L, x + np.float32(0.5), z + np.float32(2.0), INF, sig, 500.0, None, 0, B)
    assert loss_t == pytest.approx(loss, rel=1e-4)
    np.testing.assert_allclose(g_t, g, rtol=0, atol=1e-3 * np.abs(g).max())
    assert np.abs(g.sum(0)).max() < 1e-4 * np.abs(g).sum(0).max()
    # oracle on a 512-row subset, exact
    sub = slice(0, 512)
    lo_o, g_o = D.sketchmap_fwd_bwd(x[sub], z[sub], INF, sig, 500.0)
    lo, _, gg = run_sketchmap(L, x[sub], z[sub], INF, sig, 500.0, None, 0, 512)
    assert lo == pytest.approx(lo_o, rel=1e-5)
    np.testing.assert_allclose(gg, g_o, rtol=0, atol=2e-5 * np.abs(g_o).max())


@pytest.mark.parametrize("periodicity", [INF, 2 * PI, 360.0])
def test_pairwise_dist_materialised(L, periodicity):
    rng = np.random.default_rng(5)
    x = (rng.uniform(-1, 1, (70, 9)) * (PI if periodicity == INF else periodicity / 2)).astype(np.float32)
    out = torch.empty(70, 70, device="cuda")
    L.pairwise_dist(dev(x), periodicity, out)
    ref = D.pairwise_dist_direct(x) if periodicity == INF else D.pairwise_dist_periodic_direct(x, periodicity)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-6, atol=1e-5)
    # literal two-point answers of the reference tests (tests/test_pairwise_distances.py:140-160)
    pts = dev(np.array([[1 / 8, 1 / 2], [7 / 8, 1 / 2]], np.float32))
    o2 = torch.empty(2, 2, device="cuda")
    L.pairwise_dist(pts, 1.0, o2)
    np.testing.assert_allclose(o2.cpu().numpy(), [[0, 0.25], [0.25, 0]], atol=1e-6)
    L.pairwise_dist(pts, INF, o2)
    np.testing.assert_allclose(o2.cpu().numpy(), [[0, 0.75], [0.75, 0]], atol=1e-6)


@pytest.mark.parametrize("periodicity", [2 * PI, 360.0])
def test_gather_and_periodic_transforms(L, periodicity):
    rng = np.random.default_rng(6)
    N, Dd, B = 1000, 37, 129
    data = rng.uniform(-periodicity / 2, periodicity / 2, (N, Dd)).astype(np.float32)
    idx = rng.integers(0, N, B)
    xb = torch.empty(B, Dd, device="cuda")
    xin = torch.empty(B, 2 * Dd, device="cuda")
    L.gather_rows(dev(data), dev(idx, torch.int64), periodicity, xb, xin)
    assert np.array_equal(xb.cpu().numpy(), data[idx])                       # gather is bit-exact
    np.testing.assert_allclose(xin.cpu().numpy(), D.periodic_in(data[idx], periodicity), atol=2e-6)
    xin2 = torch.empty(B, 2 * Dd, device="cuda")
    L.periodic_in(xb, periodicity, xin2)
    assert torch.equal(xin, xin2)
    y = rng.normal(0, 1, (B, 2 * Dd)).astype(np.float32)
    out = torch.empty(B, Dd, device="cuda")
    L.periodic_out(dev(y), Dd, 0, Dd, periodicity, out)
    np.testing.assert_allclose(out.cpu().numpy(), D.periodic_out(y, periodicity), atol=2e-6 * periodicity)
    # gather only (non-periodic data): xin absent
    xb2 = torch.empty(B, Dd, device="cuda")
    L.gather_rows(dev(data), dev(idx, torch.int64), INF, xb2, None)
    assert torch.equal(xb, xb2)


@pytest.mark.parametrize("variant", ["mean_abs", "mean_square", "mean_norm"])
@pytest.mark.parametrize("periodicity,periodic_output", [(2 * PI, True), (360.0, True), (INF, False), (2 * PI, False)])
# 45 columns: a wave per row; 1500 columns of 131 rows: three workgroups per row; 200 columns of 300 rows: a workgroup per row;
# 1100 columns of 1500 rows: a workgroup per row, more rows than workgroups
@pytest.mark.parametrize("B,n", [(203, 45), (131, 1500), (300, 200), (1500, 1100)])
def test_periodic_cost(L, variant, periodicity, periodic_output, B, n):
    rng = np.random.default_rng(8)
    lim = PI if periodicity == INF else periodicity / 2
    target = rng.uniform(-lim, lim, (B, n)).astype(np.float32)
    width = 2 * n + 3 if periodic_output else n + 3
    y = rng.normal(0, 1, (B, width)).astype(np.float32)
    if not periodic_output:
        y[:, 1:1 + n] = rng.uniform(-lim, lim, (B, n))
    off_sin, off_cos = (1, 1 + n) if periodic_output else (1, 0)
    coef = 1.7
    for rb, re in [(0, B), (50, 120)]:
        Y = dev(y)
        dY = torch.full_like(Y, 7.0)
        loss = torch.zeros(1, device="cuda")
        ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
        L.periodic_cost_fwd_bwd(Y, off_sin, off_cos, periodic_output, dev(target), n, periodicity, variant, coef, rb, re, loss, dY, ws)
        # oracle: float64 chain rule through atan2
        y64 = y.astype(np.float64)
        if periodic_output:
            s, c = y64[:, off_sin:off_sin + n], y64[:, off_cos:off_cos + n]
            sc = periodicity / (2 * PI)
            pred = np.arctan2(s, c) * sc
        else:
            pred = y64[:, off_sin:off_sin + n]
        lo_full, g_full = D.periodic_cost_fwd_bwd(target[rb:re], pred[rb:re], periodicity, variant, coef)
        frac = (re - rb) / B
        assert float(loss) == pytest.approx(lo_full * frac, rel=2e-5, abs=1e-8)
        g_pred = g_full * frac
        got = dY.cpu().numpy().astype(np.float64)
        if periodic_output:
            r2 = s[rb:re] ** 2 + c[rb:re] ** 2
            np.testing.assert_allclose(got[rb:re, off_sin:off_sin + n], g_pred * c[rb:re] / r2 * sc, rtol=1e-4, atol=5e-8)
            np.testing.assert_allclose(got[rb:re, off_cos:off_cos + n], -g_pred * s[rb:re] / r2 * sc, rtol=1e-4, atol=5e-8)
        else:
            np.testing.assert_allclose(got[rb:re, off_sin:off_sin + n], g_pred, rtol=1e-4, atol=5e-8)
        # everything outside the addressed window / row block is untouched
        mask = np.ones_like(got, bool)
        mask[rb:re, off_sin:off_sin + n] = False
        if periodic_output:
            mask[rb:re, off_cos:off_cos + n] = False
        assert np.all(got[mask] == 7.0)


def test_center_cost(L):
    rng = np.random.default_rng(9)
    z = rng.normal(0, 2, (300, 2)).astype(np.float32)
    loss = torch.zeros(1, device="cuda")
    dZ = torch.ones(300, 2, device="cuda")
    L.center_cost_fwd_bwd(dev(z), 1e-4, 0, 300, loss, dZ, True)
    assert float(loss) == pytest.approx(1e-4 * np.mean(z.astype(np.float64) ** 2), rel=1e-5)
    np.testing.assert_allclose(dZ.cpu().numpy(), 1 + 2e-4 * z / z.size, rtol=1e-6)
    # latent == 1 -> cost == center_cost_scale (tests/test_losses.py:815-840)
    L.center_cost_fwd_bwd(torch.ones(64, 2, device="cuda"), 1e-4, 0, 64, loss, torch.zeros(64, 2, device="cuda"), False)
    assert float(loss) == pytest.approx(1e-4, rel=1e-6)


@pytest.mark.parametrize("n", [5, 1024, 66690])
def test_adam_and_tick(L, n):
    rng = np.random.default_rng(n)
    w = rng.normal(0, 0.3, n)
    l2 = (rng.random(n) < 0.9) * 2e-3
    W, M, V = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    L2 = dev(l2)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    terms = torch.zeros(4, device="cuda")
    hist = torch.zeros(8, device="cuda")
    thist = torch.zeros(8, 4, device="cuda")
    ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
    wn, m, v = w.copy(), np.zeros(n), np.zeros(n)
    for t in range(1, 6):
        g = rng.normal(0, 1.0, n) * (3.0 if t == 2 else 0.3)   # step 2 exercises the clip
        reg_o = np.sum(0.5 * l2 * wn * wn)
        L.adam_step(W, dev(g), M, V, L2, 1e-3, 0.9, 0.999, 1e-7, 1.0, step, terms[1:2], ws)
        terms[0] = float(t)
        L.step_tick(terms, 4, hist, thist, step)
        wn, m, v = D.keras_adam_step(wn, g, m, v, t, l2coef=l2)
        torch.cuda.synchronize()
        assert float(terms[1]) == pytest.approx(reg_o, rel=1e-5)
        np.testing.assert_allclose(W.cpu().numpy(), wn, rtol=0, atol=2e-6)
        assert int(step) == t
        assert float(hist[t - 1]) == pytest.approx(t + reg_o, rel=1e-5)
    assert np.allclose(thist[:5, 0].cpu().numpy(), [1, 2, 3, 4, 5])


def test_adam_after_the_bookkeeping(L):
    """``bookkeeping_done``: the optimizer launch runs AFTER the step's scalar epilogue advanced the counter, so it takes
    t from the counter as it stands and leaves the regulariser shares of the weights it has just written (the NEXT step's
    regulariser cost); emx_l2_partials produces the same shares for weights written by anything else."""
    rng = np.random.default_rng(11)
    n = 33412
    w = rng.normal(0, 0.3, n).astype(np.float32)
    l2 = ((rng.random(n) < 0.9) * 2e-3).astype(np.float32)
    W1, M1, V1 = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    W2, M2, V2 = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    L2 = dev(l2)
    s1 = torch.zeros(1, dtype=torch.int64, device="cuda")
    s2 = torch.zeros(1, dtype=torch.int64, device="cuda")
    ws1, ws2, ws3 = (torch.zeros(L.EMX_REDUCE_WS, device="cuda") for _ in range(3))
    reg = torch.zeros(4, device="cuda")
    for t in range(1, 5):
        g = dev(rng.normal(0, 1.0, n) * (3.0 if t == 2 else 0.3))
        L.adam_step(W1, g, M1, V1, L2, 1e-3, 0.9, 0.999, 1e-7, 1.0, s1, None, ws1)       # t = counter + 1 ...
        s1 += 1                                                                           # ... the epilogue counts afterwards
        s2 += 1                                                                           # here it already has
        L.adam_step(W2, g, M2, V2, L2, 1e-3, 0.9, 0.999, 1e-7, 1.0, s2, None, ws2, bookkeeping_done=True)
        L.l2_partials(W2, L2, ws3)
        torch.cuda.synchronize()
        assert int(s1) == t and int(s2) == t
        assert torch.equal(W1, W2) and torch.equal(M1, M2) and torch.equal(V1, V2)
        assert float(ws2.double().sum()) == pytest.approx(float(ws3.double().sum()), rel=1e-6)   # (the slots partition differently)
        want = float((0.5 * L2.double() * W2.double() ** 2).sum())
        L.finish_step([(ws2, L.EMX_REDUCE_WS, 1.0, None, 0, False)], reg, 4)
        assert float(reg[0]) == pytest.approx(want, rel=1e-5)


def test_adam_tick_fused_equals_separate(L):
    """emx_adam_tick (one launch: slice sum + clip + L2 + Adam + regulariser cost + bookkeeping, behind one entry
    point) gives the same result as emx_adam_step + emx_step_tick on the reduced gradient."""
    rng = np.random.default_rng(3)
    n, S = 66692, 32
    w = rng.normal(0, 0.3, n).astype(np.float32)
    l2 = ((rng.random(n) < 0.9) * 2e-3).astype(np.float32)
    parts = rng.normal(0, 0.05, (S, n)).astype(np.float32)
    P = dev(parts)
    # reference path: fixed-order slice sum as mlp_wgrad_reduce does it (8 accumulators), then the separate kernels
    W1, M1, V1 = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    W2, M2, V2 = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    L2 = dev(l2)
    s1 = torch.zeros(1, dtype=torch.int64, device="cuda"); h1 = torch.zeros(1, dtype=torch.int64, device="cuda")
    s2 = torch.zeros(1, dtype=torch.int64, device="cuda"); h2 = torch.zeros(1, dtype=torch.int64, device="cuda")
    cur = torch.zeros(1, dtype=torch.int64, device="cuda")
    t1 = torch.tensor([1.0, 0.0, 2.0, 3.0], device="cuda"); t2 = t1.clone()
    hist1 = torch.zeros(8, device="cuda"); hist2 = torch.zeros(8, device="cuda")
    th1 = torch.zeros(8, 4, device="cuda"); th2 = torch.zeros(8, 4, device="cuda")
    ws1 = torch.zeros(L.EMX_REDUCE_WS, device="cuda"); ws2 = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
    for step in range(3):
        g = P.double().sum(0).float()          # float64 sum: the fused kernel's float32 slice sum must agree to rounding
        L.adam_step(W1, g, M1, V1, L2, 1e-3, 0.9, 0.999, 1e-7, 1.0, s1, t1[1:2], ws1)
        L.step_tick(t1, 4, hist1, th1, h1, None, s1)
        L.adam_tick(W2, None, P, S, M2, V2, L2, 1e-3, 0.9, 0.999, 1e-7, 1.0, s2, t2, 4, 1, hist2, th2, h2, cur, ws2)
        torch.cuda.synchronize()
        diff = np.abs(W2.cpu().numpy() - W1.cpu().numpy())     # Adam amplifies rounding of near-zero gradients to ~lr
        assert (diff > 3e-6).mean() < 2e-3 and diff.max() < 2.1e-3
        assert int(s2) == step + 1 and int(h2) == step + 1 and int(cur) == step + 1
        assert float(t2[1]) == pytest.approx(float(t1[1]), rel=1e-5)
        assert float(hist2[step]) == pytest.approx(float(hist1[step]), rel=1e-5)
    # with the gradient given directly the two paths are bitwise identical
    W3, M3, V3 = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    W4, M4, V4 = dev(w), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    s3 = torch.zeros(1, dtype=torch.int64, device="cuda"); s4 = torch.zeros(1, dtype=torch.int64, device="cuda")
    h4 = torch.zeros(1, dtype=torch.int64, device="cuda")
    g = dev(rng.normal(0, 1, n))
    L.adam_step(W3, g, M3, V3, L2, 1e-3, 0.9, 0.999, 1e-7, 1.0, s3, t1[1:2], ws1)
    L.adam_tick(W4, g, None, 0, M4, V4, L2, 1e-3, 0.9, 0.999, 1e-7, 1.0, s4, t2, 4, 1, hist2, th2, h4, None, ws2)
    torch.cuda.synchronize()
    assert torch.equal(W3, W4) and torch.equal(M3, M4) and torch.equal(V3, V4) and float(t1[1]) == float(t2[1])


@pytest.mark.parametrize("B,Dd,periodicity,sig", [
    (8192, 512, 2 * PI, (28.0, 12, 6, 1, 2, 6)),      # BASELINE config C4 batch shape (periodic dihedrals)
    (16384, 256, INF, (22.6, 12, 6, 1, 2, 6)),        # BASELINE config C5 batch shape (Euclidean stress)
])
def test_sketchmap_baseline_stress_sizes(L, B, Dd, periodicity, sig):
    """The large BASELINE batch shapes through size-independent properties: row-block partials (the data-parallel
    rectangles) add up to the symmetric full-batch launch, the gradient has zero net force, the result is bitwise
    reproducible, and a 384-row sub-batch agrees with the float64 oracle."""
    rng = np.random.default_rng(B)
    x, z = make_xz(rng, B, Dd, 2, periodicity, 1.0)
    loss, _, g = run_sketchmap(L, x, z, periodicity, sig, 500.0, None, 0, B)
    loss2, _, g2 = run_sketchmap(L, x, z, periodicity, sig, 500.0, None, 0, B)
    assert loss == loss2 and np.array_equal(g, g2)
    nblk = 4
    step = B // nblk
    parts = [run_sketchmap(L, x, z, periodicity, sig, 500.0, None, rb, rb + step) for rb in range(0, B, step)]
    assert sum(p[0] for p in parts) == pytest.approx(loss, rel=5e-6)
    g_parts = np.concatenate([p[2][i * step:(i + 1) * step] for i, p in enumerate(parts)])
    np.testing.assert_allclose(g_parts, g, rtol=0, atol=2e-5 * np.abs(g).max())
    assert np.abs(g.sum(0)).max() < 1e-4 * np.abs(g).sum(0).max()
    sub = slice(100, 484)
    lo_o, g_o = D.sketchmap_fwd_bwd(x[sub], z[sub], periodicity, sig, 500.0)
    lo, _, gg = run_sketchmap(L, x[sub], z[sub], periodicity, sig, 500.0, None, 0, 384)
    assert lo == pytest.approx(lo_o, rel=1e-5)
    np.testing.assert_allclose(gg, g_o, rtol=0, atol=2e-5 * np.abs(g_o).max())


@pytest.mark.gpu
def test_finish_step_sums_and_bookkeeping(L):
    """emx_finish_step: every job's fixed-order sum (plain / squares, aligned / unaligned sources, lengths around the
    16-byte and the one-round boundaries, device-side scale) lands in its slot; the bookkeeping then records the total
    of ALL terms and advances the three counters -- against float64 sums."""
    g = torch.Generator().manual_seed(3)
    n_terms = 8
    lens = [1, 3, 4, 255, 2080, 8192, 8193 + 4096, 0]
    base = torch.randn(sum(lens) + 64, generator=g).cuda()
    dsc = torch.tensor([0.25], device="cuda")
    for n_jobs in (0, 1, 3, 4, 5, 8):
        terms = torch.arange(1, n_terms + 1, dtype=torch.float32, device="cuda") * 0.5
        want = terms.double().clone()
        jobs, off = [], 1   # off = 1: the first source is not 16-byte aligned
        for j in range(n_jobs):
            n = lens[j]
            src = base[off: off + n]
            sq = j % 2 == 1
            scale = 0.5 + j
            dev = dsc if j == 2 else None
            jobs.append((src.data_ptr(), n, scale, dev, j, sq))
            v = src.double()
            want[j] = float(((v * v) if sq else v).sum()) * scale * (0.25 if j == 2 else 1.0)
            off += n + (3 if j % 3 == 0 else 4 - (off + n) % 4)   # mixes aligned and unaligned starts
        hist = torch.zeros(16, device="cuda")
        thist = torch.zeros(16, n_terms, device="cuda")
        hstep = torch.tensor([5], dtype=torch.int64, device="cuda")
        cursor = torch.tensor([2], dtype=torch.int64, device="cuda")
        counter = torch.tensor([9], dtype=torch.int64, device="cuda")
        L.finish_step(jobs, terms, n_terms, hist, thist, hstep, cursor, counter)
        got = terms.double().cpu().numpy()
        scale_abs = np.maximum(1.0, np.abs(want.cpu().numpy()))
        assert np.all(np.abs(got - want.cpu().numpy()) <= 2e-5 * scale_abs * 30), (n_jobs, got, want)
        assert (int(hstep), int(cursor), int(counter)) == (6, 3, 10)
        assert float(hist[5]) == pytest.approx(float(terms.double().sum()), rel=1e-6)
        assert torch.equal(thist[5], terms)
        # sums only (no bookkeeping): counters untouched
        L.finish_step(jobs, terms, n_terms)
        assert (int(hstep), int(cursor), int(counter)) == (6, 3, 10)


def split_planes(ws, B, Dd):
    """(h + l) / s per element, norms and 1/s of an emx_sketchmap_split workspace (include/emx.h)."""
    ld = (Dd + 31) // 32 * 32
    raw = ws.cpu().numpy()
    planes = raw[: 2 * B * ld * 2].view(np.float16).reshape(2, ld // 32, B, 32).transpose(0, 2, 1, 3).reshape(2, B, ld)   # chunk-major
    tail = raw[2 * B * ld * 2: 2 * B * ld * 2 + 8 * B].view(np.float32)
    return planes.astype(np.float64), tail[:B], tail[B:], ld


@pytest.mark.gpu
@pytest.mark.parametrize("B,Dd", [(64, 128), (257, 33), (100, 7), (5, 1)])
def test_split_rows_reproduce_the_rows(L, B, Dd):
    """emx_sketchmap_split and the split made beside emx_gather_rows: the two f16 planes, scaled back, reproduce
    float32(x - x_0) to float32 accuracy whatever the magnitude of the data (per-row power-of-two scale, no f16
    overflow), the padding columns are zero, and the norms are the squared lengths of those rows."""
    rng = np.random.default_rng(B + Dd)
    x = (rng.normal(0, 3.0, (B, Dd)) * np.exp(rng.normal(0, 4.0, (B, 1)))).astype(np.float32)   # row scales over ~7 decades
    ws = torch.zeros(L.sketchmap_split_bytes(B, Dd), dtype=torch.uint8, device="cuda")
    L.sketchmap_split(dev(x), ws)

    def check(ws, rows):
        planes, norms, rscale, ld = split_planes(ws, B, Dd)
        shifted = (rows - rows[0:1]).astype(np.float32).astype(np.float64)
        assert np.all(np.isfinite(planes)) and np.all(np.abs(planes[0]) <= 2.0 ** 14 + 1)
        assert np.all(np.log2(rscale) == np.round(np.log2(rscale)))          # powers of two
        rec = planes.sum(0)[:, :Dd] * rscale[:, None].astype(np.float64)
        rowmax = np.abs(shifted).max(1, keepdims=True)
        assert np.all(np.abs(rec - shifted) <= 2.0 ** -23 * np.abs(shifted) + 2.0 ** -38 * rowmax)
        assert np.all(planes[:, :, Dd:] == 0)
        np.testing.assert_allclose(norms, (shifted ** 2).sum(1), rtol=2e-6)

    check(ws, x)
    idx = rng.permutation(B).astype(np.int64)
    ws2 = torch.zeros_like(ws)
    xb = torch.zeros(B, Dd, device="cuda")
    L.gather_rows(dev(x), dev(idx, torch.int64), INF, xb, None, split_ws=ws2)
    check(ws2, x[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("B,Dd,Ld,sig,rb,re", [
    (256, 51, 2, DEFAULT_SIG, 0, 256), (257, 33, 2, (6.0, 12, 6, 1, 2, 6), 0, 257), (1000, 128, 2, (16.0, 12, 6, 1, 2, 6), 0, 1000),
    (300, 7, 3, (2.0, 3, 6, 1, 2, 6), 0, 300), (130, 4, 4, (1.5, 12, 6, 1, 2, 6), 0, 130), (1030, 16, 2, (8.0, 12, 6, 1, 2, 6), 515, 1030),
    (4096, 128, 2, (16.0, 12, 6, 1, 2, 6), 0, 4096),
    # few tiles, many features: the reduction of a tile is split over workgroups (ADC: pair-distance vectors as points)
    (1100, 40, 2, (10.0, 12, 6, 1, 2, 6), 0, 1100),      # 18 tile columns: ragged last group of the XCD-aware tile order
    (256, 2850, 2, (80.0, 10, 5, 1, 2, 5), 0, 256), (200, 1000, 3, (45.0, 12, 6, 1, 2, 6), 0, 200), (320, 700, 2, (40.0, 12, 6, 1, 2, 6), 160, 320),
    (600, 64, 6, (8.0, 12, 6, 1, 2, 6), 0, 600), (300, 48, 8, (5.0, 3, 6, 1, 2, 6), 100, 300),       # latent widths 5 .. 8
    # batches of 8192 rows and more: 128 x 128 macro-tiles writing the 64-tile layout (sk_tile_f16_big_kernel); ragged last tiles,
    # an odd number of 64-row blocks, the general sigmoid
    (600, 64, 12, (8.0, 12, 6, 1, 2, 6), 0, 600), (300, 48, 19, (5.0, 3, 6, 1, 2, 6), 100, 300),      # latent widths beyond 8
    # latent widths 9 .. 16 on the Gram tiles (round 4: the epilogue walks the latent eight columns at a time, sk_tile_f16_kernel<16, ..>):
    # fast and general sigmoid, a row range, ragged tiles, D of one / four / eight chunks
    (700, 64, 9, (8.0, 12, 6, 1, 2, 6), 0, 700), (500, 32, 16, (5.0, 3, 6, 1, 2, 6), 200, 500), (1100, 128, 13, (16.0, 12, 6, 1, 2, 6), 0, 1100),
    (330, 256, 10, (22.0, 12, 6, 1, 2, 6), 0, 330),
    (700, 128, 1, (16.0, 12, 6, 1, 2, 6), 0, 700), (260, 256, 1, (22.0, 12, 6, 1, 2, 6), 0, 260),     # a one-column latent (compile-time width)
    (1100, 128, 3, (16.0, 12, 6, 1, 2, 6), 0, 1100), (700, 256, 4, (22.0, 12, 6, 1, 2, 6), 0, 700), (900, 128, 4, (16.0, 12, 6, 1, 2, 6), 0, 900),   # 3 / 4 columns at compile time
    (8192, 32, 2, (8.0, 12, 6, 1, 2, 6), 0, 8192), (8300, 64, 2, (11.0, 12, 6, 1, 2, 6), 0, 8300), (8256, 40, 2, (9.0, 3, 6, 1, 2, 6), 0, 8256)])
def test_sketchmap_on_split_rows(L, B, Dd, Ld, sig, rb, re):
    """The all-pairs pass with its Gram products on the 16-bit matrix cores (split rows) against the float64 direct
    formulas -- same tolerances as the float32 path -- on data far from the origin, where the row-0 shift and the
    accuracy of the split matter."""
    rng = np.random.default_rng(B + Dd)
    spread = sig[0] / math.sqrt(2 * Dd)
    x, z = make_xz(rng, B, Dd, Ld, INF, spread)
    x += 40.0 * spread          # far from the origin
    x[2] = x[3]
    z[5] = z[7]
    X, Z = dev(x), dev(z)
    split = torch.zeros(L.sketchmap_split_bytes(B, Dd), dtype=torch.uint8, device="cuda")
    L.sketchmap_split(X, split)
    ws = torch.zeros(L.sketchmap_ws_bytes(B, Ld) // 4, dtype=torch.float32, device="cuda")
    dist, cen = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    dZ = torch.full((B, Ld), float("nan"), device="cuda")
    L.sketchmap_tiles(X, Z, INF, sig, rb, re, ws, split)
    L.sketchmap_finish(B, Dd, Z, sig, 500.0, 1e-4, rb, re, dist, cen, dZ, ws)
    loss_o, g_o = D.sketchmap_fwd_bwd(x, z, INF, sig, 500.0, rb, re)
    g_o = g_o[rb:re] + 2e-4 * z[rb:re].astype(np.float64) / z.size
    g = dZ.cpu().numpy().astype(np.float64)[rb:re]
    assert float(dist) == pytest.approx(loss_o, rel=1e-5, abs=1e-9)
    np.testing.assert_allclose(g, g_o, rtol=0, atol=2e-5 * np.abs(g_o).max() + 1e-12)


def test_scratch_of_the_all_pairs_pass_needs_no_initialisation(L):
    """The split reduction of the all-pairs pass hands a tile's Gram shares to the last of its workgroups through ticket words.
    They used to live in the caller's workspace ("zero-initialised once by the caller").  Now the library keeps them in memory of
    its own, one zeroed block per workspace address (include/emx.h): a workspace FULL OF GARBAGE -- also one whose address a
    caching allocator has handed out before, zeroed then -- gives the same bits as a zeroed one, call after call; and the first
    call for a workspace inside a stream capture is refused (the block is allocated at first sight), not silently wrong."""
    from encodermap_amd import _lib

    B, Dd, Ld, sig = 256, 2850, 2, (80.0, 10, 5, 1, 2, 5)      # the ADC model's pair-distance rows: 10 tiles x 16 workgroups
    rng = np.random.default_rng(3)
    x, z = make_xz(rng, B, Dd, Ld, INF, sig[0] / math.sqrt(2 * Dd))
    X, Z = dev(x), dev(z)
    split = torch.zeros(L.sketchmap_split_bytes(B, Dd), dtype=torch.uint8, device="cuda")
    L.sketchmap_split(X, split)
    n_ws = L.sketchmap_ws_bytes(B, Ld) // 4

    def run(ws):
        dist, cen = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        dZ = torch.full((B, Ld), float("nan"), device="cuda")
        L.sketchmap_tiles(X, Z, INF, sig, 0, B, ws, split)
        L.sketchmap_finish(B, Dd, Z, sig, 500.0, 1e-4, 0, B, dist, cen, dZ, ws)
        return float(dist), dZ.cpu().numpy()

    clean = run(torch.zeros(n_ws, dtype=torch.float32, device="cuda"))      # (freed here: the allocator hands the address out again)
    loss_o, _ = D.sketchmap_fwd_bwd(x, z, INF, sig, 500.0, 0, B)
    assert clean[0] == pytest.approx(loss_o, rel=1e-5)
    dirty_ws = torch.randint(1, 2 ** 31 - 1, (n_ws,), dtype=torch.int32, device="cuda").view(torch.float32)
    for _ in range(3):
        got = run(dirty_ws)
        assert got[0] == clean[0] and np.array_equal(got[1], clean[1])
        dirty_ws.view(torch.int32).fill_(0x7fc12345)                         # garbage again between the calls (NaN patterns)
    # first sight inside a capture: refused
    keep = [torch.empty(n_ws, dtype=torch.float32, device="cuda") for _ in range(4)]      # (addresses the library has not seen)
    fresh = torch.randint(1, 1000, (n_ws,), dtype=torch.int32, device="cuda").view(torch.float32)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    dummy = torch.zeros(4, device="cuda")
    refused = None
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            dummy.add_(1.0)
            try:
                L.sketchmap_tiles(X, Z, INF, sig, 0, B, fresh, split)
                refused = False
            except _lib.EmxError as exc:
                refused = "capturing" in str(exc)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert refused is True
    del keep
    # emx_forget_counters(workspace): the address's ticket block goes back to the library's pool and the next call gets words zeroed
    # anew -- how a caller recovers after an aborted launch, and what keeps a process that creates many engines from accumulating
    # blocks; many workspaces one after the other reuse the pooled block (same results throughout)
    lib = _lib.load()
    lib.emx_forget_counters(dirty_ws.data_ptr())
    got = run(dirty_ws)
    assert got[0] == clean[0] and np.array_equal(got[1], clean[1])
    for _ in range(6):
        w = torch.randint(1, 2 ** 31 - 1, (n_ws + 64,), dtype=torch.int32, device="cuda").view(torch.float32)
        got = run(w)
        assert got[0] == clean[0] and np.array_equal(got[1], clean[1])
        torch.cuda.synchronize()
        lib.emx_forget_counters(w.data_ptr())
    lib.emx_forget_counters(None)        # (forget everything: the next call starts over)
    got = run(dirty_ws)
    assert got[0] == clean[0] and np.array_equal(got[1], clean[1])


def test_periodic_cost_batch_equals_separate_calls(L):
    """emx_periodic_cost_batch == one emx_periodic_cost_fwd_bwd per job (partials, gradients: bit for bit), jobs of different
    widths, variants and modes (atan2 output with a joined gradient, broadcast prediction without a gradient, long rows)."""
    rng = np.random.default_rng(23)
    B, W = 200, 2 * (150 + 40)
    Y = dev(rng.normal(0, 1, (B, W)))
    t1, t2 = dev(rng.uniform(-PI, PI, (B, 150))), dev(rng.uniform(-PI, PI, (B, 40)))
    add = dev(rng.normal(0, 1e-3, (B, 150)))
    mean = dev(rng.uniform(-PI, PI, (1, 60)))
    t3 = dev(rng.uniform(-PI, PI, (B, 60)))
    yl, tl = dev(rng.normal(0, 1, (B, 1500))), dev(rng.normal(0, 1, (B, 1500)))

    class Row0:   # a single row presented with row stride 0 (as the engine's _Bcast does)
        def __init__(self, t):
            self.t, self.is_cuda, self.shape = t, True, (1, t.shape[-1])

        def stride(self, i):
            return 0 if i == 0 else 1

        def data_ptr(self):
            return self.t.data_ptr()

        def is_contiguous(self):
            return True

    def jobs(dY, dYl, ws):
        return [dict(y=Y, off_sin=0, off_cos=150, periodic_output=True, target=t1, n=150, periodicity=2 * PI, variant="mean_abs", coef=0.7,
                     dY=dY, ws=ws[0], dpred_add=add),
                dict(y=Y, off_sin=300, off_cos=340, periodic_output=True, target=t2, n=40, periodicity=2 * PI, variant="mean_square", coef=1.3,
                     dY=dY, ws=ws[1]),
                dict(y=Row0(mean), off_sin=0, off_cos=0, periodic_output=False, target=t3, n=60, periodicity=2 * PI, variant="mean_abs",
                     coef=0.5, dY=None, ws=ws[2]),
                dict(y=yl, off_sin=0, off_cos=0, periodic_output=False, target=tl, n=1500, periodicity=INF, variant="mean_abs", coef=2.0,
                     dY=dYl, ws=ws[3])]

    def fresh():
        return (torch.full((B, W), float("nan"), device="cuda"), torch.full((B, 1500), float("nan"), device="cuda"),
                torch.full((4, L.EMX_REDUCE_WS), float("nan"), device="cuda"))

    dYa, dYla, wsa = fresh()
    L.periodic_cost_batch(jobs(dYa, dYla, wsa), B, 0, B)
    dYb, dYlb, wsb = fresh()
    for j in jobs(dYb, dYlb, wsb):
        L.periodic_cost_fwd_bwd(j["y"], j["off_sin"], j["off_cos"], j["periodic_output"], j["target"], j["n"], j["periodicity"], j["variant"],
                                j["coef"], 0, B, None, j["dY"], j["ws"], B=B, dpred_add=j.get("dpred_add"))
    torch.cuda.synchronize()
    assert torch.equal(wsa, wsb) and not torch.isnan(wsa).any()
    assert torch.equal(dYa, dYb) and torch.equal(dYla, dYlb) and not torch.isnan(dYa).any()
    with pytest.raises(L.EmxError, match="share their scratch"):
        js = jobs(dYa, dYla, wsa)
        js[1]["ws"] = js[0]["ws"]
        L.periodic_cost_batch(js, B, 0, B)


@pytest.mark.parametrize("B,periodicity", [(200, 2 * PI), (513, 2 * PI), (200, INF)])
def test_sketchmap_tiles_then_finish_equals_the_single_call(L, B, periodicity):
    """emx_sketchmap_tiles + emx_sketchmap_finish (given the same periodicity: it selects the tile size -- 32 x 32 for periodic
    batches of up to 512 rows -- and with it the workspace layout) == emx_sketchmap_fwd_bwd, bit for bit."""
    rng = np.random.default_rng(B)
    x, z = make_xz(rng, B, 12, 2, periodicity, 1.0)
    sig = DEFAULT_SIG if periodicity != INF else (3.0, 12, 6, 1, 2, 6)
    loss, cen, g = run_sketchmap(L, x, z, periodicity, sig, 500.0, 1e-4, 0, B)
    X, Z = dev(x), dev(z)
    ws = torch.zeros(L.sketchmap_ws_bytes(B, 2) // 4, dtype=torch.float32, device="cuda")
    dist, c2 = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    dZ = torch.full((B, 2), float("nan"), device="cuda")
    L.sketchmap_tiles(X, Z, periodicity, sig, 0, B, ws, None)
    L.sketchmap_finish(B, 12, Z, sig, 500.0, 1e-4, 0, B, dist, c2, dZ, ws, periodicity=periodicity)
    torch.cuda.synchronize()
    assert float(dist) == loss and float(c2) == cen
    assert np.array_equal(dZ.cpu().numpy().astype(np.float64), g)


@pytest.mark.parametrize("B,Dd,Ld,periodicity,tile,split", [
    (300, 12, 2, 2 * PI, 32, False),      # VALU tiles at 32 x 32 (periodic, small batch), ragged edge
    (1000, 24, 2, 2 * PI, 64, False),     # VALU tiles at 64 x 64
    (1000, 24, 2, INF, 64, True),         # f16 matrix-core tiles on split rows
    (1000, 24, 2, INF, 64, False),        # Euclidean VALU tiles
    (400, 10, 6, INF, 32, False),         # wide latent
])
@pytest.mark.parametrize("world", [2, 3])
def test_sketchmap_dealt_tiles_sum_to_the_full_pass(L, B, Dd, Ld, periodicity, tile, split, world):
    """Row-block data parallelism: the upper-triangle tiles of the symmetric pass are dealt round-robin;
    each rank's emx_sketchmap_tiles_dealt + emx_sketchmap_reduce_dealt is its share of the cost and of d cost / d z of ALL
    rows (center gradient on its own rows) -- the oracle's share for the same tiles, and summed over ranks the single call."""
    from encodermap_amd.engine import DataParallel

    rng = np.random.default_rng(B + world)
    x, z = make_xz(rng, B, Dd, Ld, periodicity, 1.0)
    sig = DEFAULT_SIG if periodicity != INF else (2.0 * math.sqrt(Dd), 12, 6, 1, 2, 6)
    loss_full, cen_full, g_full = run_sketchmap(L, x, z, periodicity, sig, 500.0, 1e-4, 0, B)
    X, Z = dev(x), dev(z)
    ws = torch.zeros(L.sketchmap_ws_bytes(B, Ld) // 4, dtype=torch.float32, device="cuda")
    split_ws = None
    if split:
        split_ws = torch.empty(L.sketchmap_split_bytes(B, Dd), dtype=torch.uint8, device="cuda")
        L.sketchmap_split(X, split_ws)
    loss_sum, g_sum = 0.0, np.zeros((B, Ld))
    for rank in range(world):
        dp = DataParallel(rank, world, "rowblock")
        rb, re = dp.row_block(B)
        ws.fill_(float("nan"))
        dZ = torch.full((B, Ld), float("nan"), device="cuda")
        L.sketchmap_tiles_dealt(X, Z, periodicity, sig, ws, split_ws, rank, world)
        L.sketchmap_reduce_dealt(B, Dd, Z, periodicity, sig, 500.0, 1e-4, rb, re, rank, world, dZ, ws)
        sk = L.sketchmap_describe(X, Z, periodicity, sig, 500.0, 1e-4, 0, B, ws)
        torch.cuda.synchronize()
        off = (sk.tile_loss - ws.data_ptr()) // 4
        tl = ws[off:off + sk.n_tiles].cpu().numpy().astype(np.float64)
        mine = dp.dealt_tiles(B, tile)
        assert sk.n_tiles == (-(-B // tile)) * (-(-B // tile) + 1) // 2
        assert np.count_nonzero(tl) == len(mine)                   # the other ranks' tiles: cost slots zeroed, not computed
        lo_o, g_o = D.sketchmap_tiles_fwd_bwd(x, z, periodicity, sig, 500.0, mine, tile)
        g_o[rb:re] += 2e-4 * z[rb:re].astype(np.float64) / z.size
        loss = tl.sum() * sk.loss_scale
        g = dZ.cpu().numpy().astype(np.float64)
        assert loss == pytest.approx(lo_o, rel=2e-5 if not split else 1e-4)
        np.testing.assert_allclose(g, g_o, rtol=0, atol=(2e-5 if not split else 2e-4) * np.abs(g_full).max())
        loss_sum += loss
        g_sum += g
    assert loss_sum == pytest.approx(loss_full, rel=2e-6 if not split else 1e-4)
    np.testing.assert_allclose(g_sum, g_full, rtol=0, atol=(1e-5 if not split else 2e-4) * np.abs(g_full).max())


@pytest.mark.parametrize("n", [1, 2, 7, 256, 1000, 100003, 1000000])
def test_permute_equals_its_host_statement(L, n):
    """emx_permute: the device shuffle is bit-identical to layout.feistel_permutation (integer work: exact), for several
    seeds and epochs, and the engine's index stream equals layout.BatchIndexStream batch for batch."""
    from encodermap_amd.layout import feistel_permutation

    for seed, epoch in ((0, 0), (5, 3), (2 ** 40 + 17, 2 ** 33 + 1)):
        out = torch.empty(n, dtype=torch.int64, device="cuda")
        L.permute(out, seed, epoch)
        assert np.array_equal(out.cpu().numpy(), feistel_permutation(n, seed, epoch)), (n, seed, epoch)


def test_engine_index_stream_equals_host_stream(L):
    import encodermap_amd as em
    from encodermap_amd.layout import BatchIndexStream

    data = np.random.default_rng(0).uniform(-3, 3, (1000, 4)).astype(np.float32)
    e = em.EncoderMap(em.Parameters(batch_size=96, seed=9), data, read_only=True)
    host = BatchIndexStream(1000, 96, seed=e._engine._shuffle_seed)   # (DataParallel.batch_seed derives the stream's seed from p.seed)
    for nb in (3, 11, 1, 25):          # 40 batches x 96 rows: almost four passes, blocks straddle them
        got = e._engine.draw_index_block(nb).cpu().numpy().reshape(nb, 96)
        assert np.array_equal(got, host.next_block(nb))
