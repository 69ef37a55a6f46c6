"""How far the engine's split-f16 arithmetic is from float64, next to how far a float32 run of the REFERENCE FORMULATION is
(oracle/reference_ops.py in torch-CPU float32) -- per quantity of one training step: activations, pre-activation gradients
(deltas), weight gradients of every Dense layer; cost and latent gradient of the all-pairs pass on data far from the origin.

Every float32 product of the dense layers and of the Euclidean all-pairs pass runs as three f16 matrix products on operands
split into two 16-bit planes (22 significant bits, csrc/wop.h, csrc/split.h): this module is the measurement behind the
claim that the result is float32-accurate.  Used by tests/test_accuracy_gpu.py (asserts engine error <= 2 x float32-oracle
error per quantity) and tools/split_accuracy.py (writes profiles/r03_split_accuracy.txt).  Test infrastructure: imports
``oracle``.
"""
import math

import numpy as np
import torch

from oracle import reference_ops as R

INF = float("inf")


def err(a, ref):
    """(rms error / rms of the reference, max error / max of the reference)"""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    d = a - ref
    return float(np.sqrt(np.mean(d * d)) / max(np.sqrt(np.mean(ref * ref)), 1e-300)), float(np.abs(d).max() / max(np.abs(ref).max(), 1e-300))


def blobs(rng, n, d, centres=16):
    c = rng.normal(0, 4, (centres, d))
    return (c[rng.integers(0, centres, n)] + rng.normal(0, 1, (n, d))).astype(np.float32)


SHAPES = {
    # BASELINE.json configs 2, 4, 5: the layer widths are what matters for the dense products; the batch is 1024 rows so that
    # the float64 oracle's [B, B(, D)] temporaries stay small (a row's numbers do not depend on the batch size, csrc/mlp.hip)
    "C2 (D=128 Euclidean)": (dict(periodicity=INF, batch_size=1024, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6)), lambda r: blobs(r, 4000, 128)),
    # (C4 with the mean_square auto cost: the default mean_abs cost differentiates to sign(delta), a DISCONTINUOUS gradient -- a
    #  handful of output elements with delta within rounding of zero flip their sign in any float32 run, which puts 1e-4 of
    #  relative error into the decoder's deltas of the float32 oracle and of the engine alike (ratios 1.4-1.8 either way, a
    #  matter of which elements flip) and says nothing about the precision of the products)
    "C4 (D=512 periodic, 1024-wide ends, mean_square)": (dict(batch_size=256, dist_sig_parameters=(30.0, 12, 6, 1, 2, 6), auto_cost_variant="mean_square"),
                                                         lambda r: r.uniform(-np.pi, np.pi, (2000, 512)).astype(np.float32)),
    "C5 (D=256 Euclidean)": (dict(periodicity=INF, batch_size=1024, dist_sig_parameters=(22.6, 12, 6, 1, 2, 6)), lambda r: blobs(r, 4000, 256)),
}


def _oracle_step(D, p, w0, x, dtype):
    """one forward + backward of the reference formulation; returns activations h_l, deltas dL/da_l, weight gradients (the
    regulariser's share excluded: the engine adds it inside the optimizer), all as float64 numpy"""
    model = R.SequentialOracle(D, p.n_neurons, p.activation_functions, p.periodicity, w0, dtype=dtype)
    hs, pres = [], []
    orig = R.dense_stack

    def capture(xx, weights, biases, acts):
        for W, b, act in zip(weights, biases, acts):
            a = xx @ W + b
            a.retain_grad()
            pres.append(a)
            xx = torch.tanh(a) if act == "tanh" else a
            hs.append(xx)
        return xx

    R.dense_stack = capture
    try:
        total, terms = R.encodermap_loss(model, torch.tensor(x, dtype=dtype), p.to_dict())
        (total - terms["reg"]).backward()
    finally:
        R.dense_stack = orig
    f = lambda t: t.detach().to(torch.float64).numpy()
    return [f(h) for h in hs], [f(a.grad) for a in pres], [f(q.grad) for q in model.params], float(total.detach())


def dense_layer_errors(em, name, seed=0, local=False):
    """{quantity: {"engine": (rms, max), "oracle_f32": (rms, max)}} for one step at the shape ``name``.
    ``local``: the weight gradients of emx_mlp_wgrad_adam_local (summed inside the workgroups, consumed by the optimizer in the same
    launch -- never stored): recovered from Adam's first moment after ONE step from zero moments, m = 0.1 (g + l2coef w) where the
    clip at 1 is not reached (elements where it is are left out)."""
    from encodermap_amd import _lib
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    kw, maker = SHAPES[name]
    rng = np.random.default_rng(seed)
    data = maker(rng)
    p = em.Parameters(n_steps=1, **kw)
    if local:      # every cost scaled down: the gradients -- whose relative error is what is measured -- stay below Adam's clip at 1
        p = em.Parameters(n_steps=1, **dict(kw, distance_cost_scale=p.distance_cost_scale * 1e-4, auto_cost_scale=p.auto_cost_scale * 1e-4,
                                            center_cost_scale=p.center_cost_scale * 1e-4, l2_reg_constant=p.l2_reg_constant * 1e-4))
    D = data.shape[1]
    net_dim = 2 * D if p.periodicity < INF else D
    lay = build_layout(p.n_neurons, p.activation_functions, net_dim)
    w0 = unpack_weights(lay, init_parameters(lay, 3))
    B = p.batch_size
    idx = rng.integers(0, data.shape[0], B)
    e = em.EncoderMap(p, data, model=[w.copy() for w in w0], read_only=True, use_graph=False)
    eng = e._engine
    eng._lean = False      # (this comparison reads every float32 row of the step: activations and deltas of all layers)
    if local:
        w_before = eng.w.double().cpu().numpy()
        eng.run_steps(1, torch.from_numpy(idx).to(eng.device))
        torch.cuda.synchronize()
        assert eng._local_ok and eng._tail_ran, "emx_mlp_wgrad_adam_local is not made for this shape"
        gt = eng.m.double().cpu().numpy() / (1.0 - 0.9)
        l2c = eng.l2coef.double().cpu().numpy() if eng.l2coef is not None else 0.0
        g = gt - l2c * w_before
        unclipped = np.abs(gt) < 0.999
    else:
        # the step up to the weight gradients (no optimizer), as run_steps would enqueue it with the two-launch tail
        eng._local_ok = False
        eng.idx_block = torch.zeros(eng._block_capacity() * B, dtype=torch.int64, device=eng.device)
        eng.idx_block[:B].copy_(torch.from_numpy(idx).to(eng.device))
        eng.cursor.zero_()
        eng._gather_current()
        eng._step_pre()
        torch.cuda.synchronize()
        n_sl = _lib.mlp_wgrad_slices(B)
        g = eng.wg_ws[: n_sl * eng.w.numel()].view(n_sl, -1).double().sum(0).cpu().numpy()
        unclipped = np.ones(g.shape, bool)
    e_h = [t.cpu().numpy() for t in eng.h]
    e_d = [t.cpu().numpy() for t in eng.delta]
    e_g = unpack_weights(eng.layout, g)
    keep = unpack_weights(eng.layout, unclipped.astype(np.float64))
    x = data[idx]
    h64, d64, g64, _ = _oracle_step(D, p, w0, x, torch.float64)
    h32, d32, g32, _ = _oracle_step(D, p, w0, x, torch.float32)
    out = {}
    nl = len(h64)
    for l in range(nl):
        out[f"h[{l}] activations {h64[l].shape[1]}"] = {"engine": err(e_h[l], h64[l]), "oracle_f32": err(h32[l], h64[l])}
    for l in range(nl):
        # (the periodic model's last delta is d loss / d output of the last Dense layer, in front of the atan2: both sides)
        out[f"delta[{l}] {d64[l].shape[1]}"] = {"engine": err(e_d[l], d64[l]), "oracle_f32": err(d32[l], d64[l])}
    for l in range(nl):
        for i, nm in ((2 * l, f"dW[{l}] {g64[2 * l].shape[0]}x{g64[2 * l].shape[1]}"), (2 * l + 1, f"db[{l}]")):
            k = keep[i] > 0.5
            assert k.mean() > 0.9, "most gradient elements should be below Adam's clip"
            out[nm] = {"engine": err(np.where(k, e_g[i], g64[i]), g64[i]), "oracle_f32": err(np.where(k, g32[i], g64[i]), g64[i])}
    return out


def all_pairs_errors(B, D, sig=None, offset_sigmas=40.0, seed=0):
    """cost and latent gradient of the Euclidean all-pairs pass on the f16 matrix cores, data ``offset_sigmas`` spreads away
    from the origin (where the Gram identity cancels and the row-0 shift of csrc/split.h matters): engine and float32 oracle
    (the reference's Gram formulation, distances.py:216-253, in float32) against the float64 oracle"""
    from encodermap_amd import _lib as L

    rng = np.random.default_rng(seed + B + D)
    sig = sig or (math.sqrt(2 * D), 12, 6, 1, 2, 6)
    spread = sig[0] / math.sqrt(2 * D)
    x = (rng.normal(0, spread, (B, D)) + offset_sigmas * spread).astype(np.float32)
    z = rng.normal(0, 1.0, (B, 2)).astype(np.float32)
    X, Z = torch.from_numpy(x).cuda(), torch.from_numpy(z).cuda()
    split = torch.zeros(L.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
    L.sketchmap_split(X, split)
    ws = torch.zeros(L.sketchmap_ws_bytes(B, 2) // 4, dtype=torch.float32, device="cuda")
    dist, cen = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
    dZ = torch.full((B, 2), float("nan"), device="cuda")
    L.sketchmap_tiles(X, Z, INF, sig, 0, B, ws, split)
    L.sketchmap_finish(B, D, Z, sig, 500.0, 0.0, 0, B, dist, cen, dZ, ws)
    res = {}
    for dt in (torch.float64, torch.float32):
        zt = torch.tensor(z, dtype=dt, requires_grad=True)
        c = R.distance_cost(torch.tensor(x, dtype=dt), zt, INF, sig, 500.0)
        c.backward()
        res[dt] = (float(c.detach()), zt.grad.to(torch.float64).numpy())
    c64, g64 = res[torch.float64]
    c32, g32 = res[torch.float32]
    ce = float(dist)
    return {"distance cost": {"engine": (abs(ce - c64) / abs(c64),) * 2, "oracle_f32": (abs(c32 - c64) / abs(c64),) * 2},
            "dZ (latent gradient)": {"engine": err(dZ.cpu().numpy(), g64), "oracle_f32": err(g32, g64)}}


def format_table(title, table):
    lines = [title, f"  {'quantity':38s} {'engine rms':>11s} {'f32 rms':>11s} {'ratio':>6s}   {'engine max':>11s} {'f32 max':>11s} {'ratio':>6s}"]
    for k, v in table.items():
        (er, em_), (orr, om) = v["engine"], v["oracle_f32"]
        lines.append(f"  {k:38s} {er:11.3e} {orr:11.3e} {er / max(orr, 1e-300):6.2f}   {em_:11.3e} {om:11.3e} {em_ / max(om, 1e-300):6.2f}")
    return "\n".join(lines)
