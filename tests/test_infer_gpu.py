"""The streaming inference launch (emx_mlp_infer: encode() / decode() / generate(), autoencoder.py:1110-1195; SequentialModel.encoder /
decoder, models.py:3331-3359) against a float64 statement of the same layers (Keras Dense, PeriodicInput layers.py:204-215,
PeriodicOutput layers.py:1118-1125).  Needs an MI355X.

Tolerances: layer outputs 5e-6 relative to max(1, largest output) (the float32 resolution of split-f16 x 3 products), angles of a periodic
output 2e-5 of a radian, times 1 / |(sin, cos)| where the pair is shorter than 1 (a rounding of the pair turns the angle by that much); encode() / decode() of the facade within the north
star's 1e-4."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
INF = float("inf")
PI = math.pi


@pytest.fixture(scope="module")
def L():
    from encodermap_amd import _lib

    _lib.require_gpu()
    _lib.load()
    return _lib


ACTS = {
    None: lambda x: x, "tanh": np.tanh, "relu": lambda x: np.maximum(x, 0), "sigmoid": lambda x: 1 / (1 + np.exp(-x)),
    "elu": lambda x: np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1), "softplus": lambda x: np.logaddexp(x, 0),
    "selu": lambda x: 1.0507009873554805 * np.where(x > 0, x, 1.6732632423543772 * (np.exp(np.minimum(x, 0)) - 1)),
    "softsign": lambda x: x / (1 + np.abs(x)), "exponential": np.exp,
}


def reference(x, arrays, acts, l0, l1, p_in, p_out):
    cur = np.asarray(x, np.float64)
    if p_in:
        cur = np.concatenate([np.sin(cur * 2 * PI / p_in), np.cos(cur * 2 * PI / p_in)], axis=1)
    for i in range(l0, l1):
        cur = ACTS[acts[i]](cur @ arrays[2 * i].astype(np.float64) + arrays[2 * i + 1].astype(np.float64))
    if p_out:
        n = cur.shape[1] // 2
        reference.radius = np.hypot(cur[:, :n], cur[:, n:])      # (how well conditioned each angle is: angle_close)
        cur = np.arctan2(cur[:, :n], cur[:, n:]) * p_out / (2 * PI)
    return cur


def angle_close(a, b, period, atol, radius=None):
    """|a - b| on the circle below atol -- scaled, where the (sin, cos) pair behind an angle is short, by 1 / its length: an error e of
    the pair's components turns the angle by e / radius"""
    d = np.abs(a - b)
    d = np.minimum(d, period - d)
    tol = atol * (np.maximum(1.0, 1.0 / radius) if radius is not None else 1.0)
    return bool((d < tol).all())


CASES = [
    # D (raw features), n_neurons, activation_functions, periodicity, rows
    (128, [128, 128, 2], ["", "tanh", "tanh", ""], INF, 4096),           # C2
    (128, [128, 128, 2], ["", "tanh", "tanh", ""], INF, 100003),         # many blocks per wave, ragged last block
    (10, [128, 128, 2], ["", "tanh", "tanh", ""], 2 * PI, 1000),         # C1, the default configuration: periodic in and out
    (51, [128, 128, 2], ["", "tanh", "tanh", ""], INF, 333),             # rows of 51 floats: scalar loads, four K-steps
    (64, [128, 128, 2], ["", "tanh", "tanh", ""], 360.0, 515),           # the widest periodic model the pass takes (2 x 64)
    (5, [64, 32, 3], ["", "tanh", "tanh", ""], 2 * PI, 77),              # odd periodic width, narrow hidden layers, latent of three
    (7, [64, 3], ["", "tanh", ""], INF, 1),                              # one row
    (33, [50, 17, 4], ["", "relu", "sigmoid", "tanh"], INF, 200),        # other activations, an activation on the latent layer
    (20, [100, 30, 2], ["", "softplus", "elu", ""], INF, 64),
    (12, [40, 2], ["", "selu", "softsign"], INF, 31),
    (16, [128, 2], ["", "exponential", ""], INF, 33),
    (100, [128, 128, 128, 2], ["", "tanh", "tanh", "tanh", ""], INF, 50),   # three 128-wide hidden layers: pieces beyond LDS -> EMX_EUNSUPPORTED
    # one WIDE end (round 6: the first / last layer's pieces staged chunk by chunk): [wide, tanh 65..128, tanh 65..128, last] shapes
    (200, [128, 128, 2], ["", "tanh", "tanh", ""], INF, 50),
    (256, [128, 128, 2], ["", "tanh", "tanh", ""], INF, 70001),             # C5's network; several passes of a workgroup, ragged end
    (512, [128, 128, 2], ["", "tanh", "tanh", ""], 2 * PI, 700),            # C4's network: 1024 sines and cosines in, 512 angles out
    (300, [100, 70, 3], ["", "tanh", "tanh", ""], INF, 333),                # narrower hidden layers, a latent of three, scalar row accesses? (300 = 4 x 75)
    (131, [128, 128, 2], ["", "tanh", "tanh", ""], 360.0, 100),             # odd periodic width: K0 = 262, scalar accesses
    (1001, [128, 96, 4], ["", "tanh", "tanh", ""], INF, 260),               # 32 K-steps, the last one ragged, rows of odd length
    (150, [128, 128, 2], ["", "relu", "tanh", ""], INF, 40),                # a wide end with another hidden activation -> EMX_EUNSUPPORTED
    (150, [64, 128, 2], ["", "tanh", "tanh", ""], INF, 40),                 # a wide end in front of a narrow hidden layer -> EMX_EUNSUPPORTED
]


def streamable(layers):
    """what emx_mlp_infer takes in one launch (csrc/infer.hip): every layer within 128 x 128 and at most three of them at full width
    (LDS), or ONE wide end on the standard shape"""
    wide = [max(l.n_in, l.n_out) > 128 for l in layers]
    if not any(wide):
        return len(layers) <= 3
    first, last = layers[0], layers[-1]
    one_end = (wide[0] and first.n_out <= 128 and not any(wide[1:])) or (wide[-1] and last.n_in <= 128 and not any(wide[:-1]))
    return (one_end and len(layers) == 3 and all(l.activation == "tanh" for l in layers[:2])
            and all(64 < l.n_out <= 128 for l in layers[:2]))


@pytest.mark.parametrize("D,n_neurons,afs,periodicity,rows", CASES)
def test_streaming_inference_against_float64(L, D, n_neurons, afs, periodicity, rows):
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    rng = np.random.default_rng(D * 131 + rows)
    periodic = periodicity != INF
    net_dim = 2 * D if periodic else D
    lay = build_layout(n_neurons, afs, net_dim)
    flat = init_parameters(lay, 3)
    arrays = unpack_weights(lay, flat)
    acts = [l.activation for l in lay.layers]
    n, ne = len(lay.layers), lay.n_enc
    params = torch.from_numpy(flat).cuda()
    plan = L.MlpPlan(lay, params, [None] * n, [None] * n)
    if periodic:
        x = rng.uniform(-periodicity / 2, periodicity / 2, (rows, D)).astype(np.float32)
    else:
        scales = [0.3, 1.0] if "exponential" in afs else [0.01, 1.0, 30.0]      # rows of very different scale (exp: within float32)
        x = (rng.normal(0, 1, (rows, D)) * rng.choice(scales, (rows, 1))).astype(np.float32)
    z = rng.normal(0, 1.5, (rows, lay.latent_dim)).astype(np.float32)
    X, Z = torch.from_numpy(x).cuda(), torch.from_numpy(z).cuda()
    enc = torch.full((rows, lay.latent_dim), float("nan"), device="cuda")
    dec = torch.full((rows, D), float("nan"), device="cuda")
    if not (streamable(lay.layers[:ne]) and streamable(lay.layers[ne:])):
        c0 = L.load().emx_launch_count()
        for l0, l1, src, dst, p_in, p_out in ((0, ne, X, enc, periodicity if periodic else 0.0, 0.0), (ne, n, Z, dec, 0.0, periodicity if periodic else 0.0)):
            if not streamable(lay.layers[l0:l1]):
                with pytest.raises(L.EmxError) as err:
                    L.mlp_infer(plan, l0, l1, src, dst, p_in, p_out)
                assert err.value.unsupported
        assert L.load().emx_launch_count() == c0                                # decided in front of the launch
        return
    c0 = L.load().emx_launch_count()
    L.mlp_infer(plan, 0, ne, X, enc, periodicity if periodic else 0.0, 0.0)
    L.mlp_infer(plan, ne, n, Z, dec, 0.0, periodicity if periodic else 0.0)
    assert L.load().emx_launch_count() - c0 == 2                                # one launch per pass
    torch.cuda.synchronize()
    enc_o = reference(x, arrays, acts, 0, ne, periodicity if periodic else None, None)
    dec_o = reference(z, arrays, acts, ne, n, None, periodicity if periodic else None)
    np.testing.assert_allclose(enc.cpu().numpy(), enc_o, rtol=5e-6, atol=5e-6 * max(1.0, np.abs(enc_o).max()))
    if periodic:
        assert angle_close(dec.cpu().numpy().astype(np.float64), dec_o, periodicity, 2e-5 * periodicity / (2 * PI), reference.radius)
    else:
        np.testing.assert_allclose(dec.cpu().numpy(), dec_o, rtol=5e-6, atol=5e-6 * max(1.0, np.abs(dec_o).max()))
    # the same rows through the training kernels' forward pass (emx_mlp_forward + emx_periodic_in): float32 rounding apart
    h = [torch.empty(rows, l.n_out, device="cuda") for l in lay.layers]
    plan2 = L.MlpPlan(lay, params, h, [None] * n, mirror=plan.mirror)
    xin = X
    if periodic:
        xin = torch.empty(rows, 2 * D, device="cuda")
        L.periodic_in(X, periodicity, xin)
    L.mlp_forward(plan2, 0, ne, xin, 0, rows, train=False)
    torch.cuda.synchronize()
    np.testing.assert_allclose(enc.cpu().numpy(), h[ne - 1].cpu().numpy(), rtol=1e-5, atol=1e-5 * max(1.0, np.abs(enc_o).max()))


def test_streaming_inference_strides_alignment_and_row_independence(L):
    """Row strides beyond the row length, base pointers that allow 16-, 8- and 4-byte accesses only, and the same rows in another
    position of another call: a row's result depends on nothing but the row (bit for bit)."""
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    rng = np.random.default_rng(11)
    lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], 128)
    flat = init_parameters(lay, 5)
    arrays = unpack_weights(lay, flat)
    acts = [l.activation for l in lay.layers]
    n, ne = len(lay.layers), lay.n_enc
    params = torch.from_numpy(flat).cuda()
    plan = L.MlpPlan(lay, params, [None] * n, [None] * n)
    rows = 700
    x = rng.normal(0, 2, (rows, 128)).astype(np.float32)
    want = reference(x, arrays, acts, 0, ne, None, None)
    base = None
    for shift, ld in ((0, 128), (0, 132), (2, 130), (1, 129), (3, 131)):
        buf = torch.zeros(rows * ld + 8, device="cuda")
        X = buf[shift: shift + rows * ld].view(rows, ld)[:, :128]
        X.copy_(torch.from_numpy(x))
        ldo = 3 if shift & 1 else 4                                               # (4: the two result columns leave as one 8-byte store)
        obuf = torch.full((rows * ldo + 4,), float("nan"), device="cuda")
        O = obuf[shift & 1: (shift & 1) + rows * ldo].view(rows, ldo)[:, :2]
        L.mlp_infer(plan, 0, ne, X, O)
        torch.cuda.synchronize()
        got = O.cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=5e-6, atol=5e-6 * np.abs(want).max())
        assert torch.isnan(obuf[(shift & 1) + 2:: ldo][: rows - 1]).all()        # the column beside the result rows is untouched
        base = got if base is None else base
        assert np.array_equal(got, base)                                          # (vector width of the accesses changes no bit)
    # rows reversed, in a call of another size
    Xr = torch.from_numpy(np.ascontiguousarray(x[::-1][:333])).cuda()
    Or = torch.empty(333, 2, device="cuda")
    L.mlp_infer(plan, 0, ne, Xr, Or)
    torch.cuda.synchronize()
    assert np.array_equal(Or.cpu().numpy(), base[::-1][:333])
    # decoder: 2 -> 128 -> 128 -> 128 with a row stride and full 16-byte stores or 8- / 4-byte ones
    z = rng.normal(0, 1, (rows, 2)).astype(np.float32)
    want = reference(z, arrays, acts, ne, n, None, None)
    for shift, ld in ((0, 128), (2, 130), (1, 129)):
        obuf = torch.zeros(rows * ld + 8, device="cuda")
        O = obuf[shift: shift + rows * ld].view(rows, ld)[:, :128]
        L.mlp_infer(plan, ne, n, torch.from_numpy(z).cuda(), O)
        torch.cuda.synchronize()
        np.testing.assert_allclose(O.cpu().numpy(), want, rtol=5e-6, atol=5e-6 * np.abs(want).max())


def test_streaming_inference_random_shapes(L):
    """Forty networks drawn at random among the shapes the pass takes -- narrow, ragged, periodic or not, one end wide or none --
    with random row counts, row strides and base alignments, each against float64.  (Every draw is reproducible: seed 2026 + i.)"""
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    for i in range(40):
        rng = np.random.default_rng(2026 + i)
        periodic = bool(rng.integers(0, 2))
        wide = bool(rng.integers(0, 2))
        if wide:
            D = int(rng.integers(129, 700)) if not periodic else int(rng.integers(65, 400))
            n_neurons = [int(rng.integers(65, 129)), int(rng.integers(65, 129)), int(rng.integers(1, 9))]
            afs = ["", "tanh", "tanh", ""]
        else:
            D = int(rng.integers(1, 129)) if not periodic else int(rng.integers(1, 65))
            n_neurons = [int(rng.integers(3, 129)), int(rng.integers(3, 129)), int(rng.integers(1, 9))]
            afs = ["", str(rng.choice(["tanh", "relu", "sigmoid"])), "tanh", str(rng.choice(["", "tanh"]))]
        rows = int(rng.integers(1, 3000))
        periodicity = float(rng.choice([2 * PI, 360.0])) if periodic else INF
        lay = build_layout(n_neurons, afs, 2 * D if periodic else D)
        if not (streamable(lay.layers[:lay.n_enc]) and streamable(lay.layers[lay.n_enc:])):
            continue
        flat = init_parameters(lay, 100 + i)
        arrays = unpack_weights(lay, flat)
        acts = [l.activation for l in lay.layers]
        n, ne = len(lay.layers), lay.n_enc
        plan = L.MlpPlan(lay, torch.from_numpy(flat).cuda(), [None] * n, [None] * n)
        x = (rng.uniform(-periodicity, periodicity, (rows, D)) if periodic else rng.normal(0, 1, (rows, D)) * rng.choice([0.02, 1.0, 20.0], (rows, 1))).astype(np.float32)
        z = rng.normal(0, 1.5, (rows, lay.latent_dim)).astype(np.float32)
        ld, shift = D + int(rng.integers(0, 4)), int(rng.integers(0, 4))        # row stride and base alignment of the input rows
        xbuf = torch.zeros(rows * ld + 8, device="cuda")
        X = xbuf[shift: shift + rows * ld].view(rows, ld)[:, :D]
        X.copy_(torch.from_numpy(x))
        ldo = D + int(rng.integers(0, 4))
        obuf = torch.full((rows * ldo + 8,), float("nan"), device="cuda")
        O = obuf[shift: shift + rows * ldo].view(rows, ldo)[:, :D]
        enc = torch.full((rows, lay.latent_dim), float("nan"), device="cuda")
        L.mlp_infer(plan, 0, ne, X, enc, periodicity if periodic else 0.0, 0.0)
        L.mlp_infer(plan, ne, n, torch.from_numpy(z).cuda(), O, 0.0, periodicity if periodic else 0.0)
        torch.cuda.synchronize()
        what = f"draw {i}: D={D} n_neurons={n_neurons} afs={afs} periodic={periodic} rows={rows} ld={ld}/{ldo} shift={shift}"
        enc_o = reference(x, arrays, acts, 0, ne, periodicity if periodic else None, None)
        dec_o = reference(z, arrays, acts, ne, n, None, periodicity if periodic else None)
        np.testing.assert_allclose(enc.cpu().numpy(), enc_o, rtol=5e-6, atol=5e-6 * max(1.0, np.abs(enc_o).max()), err_msg=what)
        if periodic:
            assert angle_close(O.cpu().numpy().astype(np.float64), dec_o, periodicity, 2e-5 * periodicity / (2 * PI), reference.radius), what
        else:
            np.testing.assert_allclose(O.cpu().numpy(), dec_o, rtol=5e-6, atol=5e-6 * max(1.0, np.abs(dec_o).max()), err_msg=what)
        if ldo > D:
            assert torch.isnan(obuf[shift + D:: ldo][: rows - 1]).all(), what       # nothing written beside the rows


def test_streaming_inference_rejects(L):
    from encodermap_amd.layout import build_layout, init_parameters

    lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], 21)
    params = torch.from_numpy(init_parameters(lay, 1)).cuda()
    plan = L.MlpPlan(lay, params, [None] * 6, [None] * 6)
    X = torch.zeros(8, 21, device="cuda")
    O = torch.zeros(8, 2, device="cuda")
    with pytest.raises(L.EmxError, match="even first width"):
        L.mlp_infer(plan, 0, 3, X, O, 2 * PI, 0.0)           # 21 input columns cannot be [sin | cos]
    with pytest.raises(L.EmxError, match="layer range"):
        L.mlp_infer(plan, 3, 3, X, O)
    L.mlp_infer(plan, 0, 3, X[:0], O[:0])                    # no rows: nothing to do, no error


@pytest.mark.parametrize("D", [12, 300])
@pytest.mark.parametrize("periodicity", [INF, 2 * PI])
def test_facade_encode_decode_use_the_streaming_launch(L, periodicity, D, tmp_path):
    """em.EncoderMap.encode() / decode() / generate() against the float64 oracle model (north star: 1e-4), one launch per call,
    and -- EMX_INFER_STREAM=0's path -- the training kernels' forward pass agreeing with it."""
    import encodermap_amd as em
    from encodermap_amd.engine import SequentialEngine
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    rng = np.random.default_rng(3)
    periodic = periodicity != INF                            # (D = 300: one wide end, the staged form of the launch)
    x = (rng.uniform(-PI, PI, (5000, D)) if periodic else rng.normal(0, 1, (5000, D))).astype(np.float32)
    p = em.Parameters(periodicity=periodicity, n_steps=1, batch_size=256, main_path=str(tmp_path))
    lay = build_layout(p.n_neurons, p.activation_functions, 2 * D if periodic else D)
    w0 = unpack_weights(lay, init_parameters(lay, 4))
    e = em.EncoderMap(p, x, model=w0, read_only=True)
    acts = [l.activation for l in lay.layers]
    c0 = L.load().emx_launch_count()
    z = e.encode(x)
    assert L.load().emx_launch_count() - c0 == 1
    z_o = reference(x, w0, acts, 0, lay.n_enc, periodicity if periodic else None, None)
    assert np.abs(z - z_o).max() < 1e-4 * max(1.0, np.abs(z_o).max())
    assert np.array_equal(e.encode(), z)                     # the resident copy: the same kernel, the same bits
    y = e.decode(z[:777])
    y_o = reference(z[:777], w0, acts, lay.n_enc, len(lay.layers), None, periodicity if periodic else None)
    radius = getattr(reference, "radius", None)              # (angles behind a short (sin, cos) pair: tolerance / length, see angle_close)
    if periodic:
        assert angle_close(y.astype(np.float64), y_o, periodicity, 1e-4, radius)
    else:
        assert np.abs(y - y_o).max() < 1e-4 * max(1.0, np.abs(y_o).max())
    assert np.array_equal(np.asarray(e.generate(z[:777])), y)
    old = SequentialEngine._infer_streaming
    try:
        SequentialEngine._infer_streaming = False
        z2, y2 = e.encode(x), e.decode(z[:777])
    finally:
        SequentialEngine._infer_streaming = old
    np.testing.assert_allclose(z2, z, rtol=1e-5, atol=1e-5)
    if periodic:
        assert angle_close(y2.astype(np.float64), y.astype(np.float64), periodicity, 1e-4, radius)
    else:
        np.testing.assert_allclose(y2, y, rtol=1e-5, atol=1e-5)
