"""Soak of the streaming inference launch: N random networks / row counts / strides against float64 (the draws of
tests/test_infer_gpu.py::test_streaming_inference_random_shapes, more of them).   python tools/infer_soak.py [N=300] [seed0=5000]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import test_infer_gpu as T
from encodermap_amd import _lib as L
from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
L.require_gpu(); L.load()
PI, INF = T.PI, T.INF
ran = skipped = 0
worst_e = worst_d = 0.0
for i in range(N):
    rng = np.random.default_rng(seed0 + i)
    periodic = bool(rng.integers(0, 2)); wide = bool(rng.integers(0, 2))
    if wide:
        D = int(rng.integers(129, 1200)) if not periodic else int(rng.integers(65, 600))
        n_neurons = [int(rng.integers(65, 129)), int(rng.integers(65, 129)), int(rng.integers(1, 9))]
        afs = ["", "tanh", "tanh", ""]
    else:
        D = int(rng.integers(1, 129)) if not periodic else int(rng.integers(1, 65))
        n_neurons = [int(rng.integers(3, 129)), int(rng.integers(3, 129)), int(rng.integers(1, 9))]
        afs = ["", str(rng.choice(["tanh", "relu", "sigmoid", "elu"])), "tanh", str(rng.choice(["", "tanh"]))]
    rows = int(rng.integers(1, 9000))
    periodicity = float(rng.choice([2 * PI, 360.0])) if periodic else INF
    lay = build_layout(n_neurons, afs, 2 * D if periodic else D)
    if not (T.streamable(lay.layers[:lay.n_enc]) and T.streamable(lay.layers[lay.n_enc:])):
        skipped += 1
        continue
    flat = init_parameters(lay, 100 + i); arrays = unpack_weights(lay, flat)
    acts = [l.activation for l in lay.layers]; n, ne = len(lay.layers), lay.n_enc
    plan = L.MlpPlan(lay, torch.from_numpy(flat).cuda(), [None] * n, [None] * n)
    x = (rng.uniform(-periodicity, periodicity, (rows, D)) if periodic else rng.normal(0, 1, (rows, D)) * rng.choice([0.02, 1.0, 20.0], (rows, 1))).astype(np.float32)
    z = rng.normal(0, 1.5, (rows, lay.latent_dim)).astype(np.float32)
    ld, shift = D + int(rng.integers(0, 4)), int(rng.integers(0, 4))
    xbuf = torch.zeros(rows * ld + 8, device="cuda"); X = xbuf[shift: shift + rows * ld].view(rows, ld)[:, :D]; X.copy_(torch.from_numpy(x))
    ldo = D + int(rng.integers(0, 4)); obuf = torch.full((rows * ldo + 8,), float("nan"), device="cuda"); O = obuf[shift: shift + rows * ldo].view(rows, ldo)[:, :D]
    enc = torch.full((rows, lay.latent_dim), float("nan"), device="cuda")
    L.mlp_infer(plan, 0, ne, X, enc, periodicity if periodic else 0.0, 0.0)
    L.mlp_infer(plan, ne, n, torch.from_numpy(z).cuda(), O, 0.0, periodicity if periodic else 0.0)
    torch.cuda.synchronize()
    what = f"draw {seed0 + i}: D={D} n_neurons={n_neurons} afs={afs} periodic={periodic} rows={rows} ld={ld}/{ldo} shift={shift}"
    enc_o = T.reference(x, arrays, acts, 0, ne, periodicity if periodic else None, None)
    dec_o = T.reference(z, arrays, acts, ne, n, None, periodicity if periodic else None)
    e = np.abs(enc.cpu().numpy() - enc_o).max() / max(1.0, np.abs(enc_o).max())
    if e >= 5e-6:
        # unbounded hidden activations on rows of scale 20: what does a float32 run of the same layers lose?
        cur = x.astype(np.float32)
        if periodic:
            cur = np.concatenate([np.sin(cur * np.float32(2 * PI / periodicity)), np.cos(cur * np.float32(2 * PI / periodicity))], axis=1).astype(np.float32)
        for li in range(ne):
            cur = T.ACTS[acts[li]]((cur @ arrays[2 * li].astype(np.float32) + arrays[2 * li + 1].astype(np.float32)).astype(np.float32)).astype(np.float32)
        e32 = np.abs(cur - enc_o).max() / max(1.0, np.abs(enc_o).max())
        print(f"{what}: error {e:.2e} beyond 5e-6; a float32 numpy run of the same layers: {e32:.2e}", flush=True)
        assert e < 4 * e32, (what, e, e32)
    worst_e = max(worst_e, e)
    if periodic:
        assert T.angle_close(O.cpu().numpy().astype(np.float64), dec_o, periodicity, 2e-5 * periodicity / (2 * PI), T.reference.radius), what
    else:
        d = np.abs(O.cpu().numpy() - dec_o).max() / max(1.0, np.abs(dec_o).max())
        assert d < 5e-6, (what, d)
        worst_d = max(worst_d, d)
    if ldo > D:
        assert torch.isnan(obuf[shift + D:: ldo][: rows - 1]).all(), what
    ran += 1
print(f"{ran} networks checked ({skipped} draws outside the pass's shapes), worst relative error encode {worst_e:.2e}, decode (non-periodic) {worst_d:.2e}")
