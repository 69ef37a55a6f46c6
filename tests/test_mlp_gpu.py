"""Fused dense-layer kernels (emx_mlp_forward / backward / wgrad: float32 operands as scaled f16 pairs on the 16-bit
matrix cores) against a float64 torch reference of the same layers (Keras Dense semantics,
encodermap/models/models.py:3179-3220).  Needs an MI355X.
Tolerances: activations 2e-6 * (1 + |x|), gradients 2e-5 relative to the largest entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from encodermap_amd import _lib

    _lib.require_gpu()
    _lib.load()
    return _lib


def reference(x, arrays, acts, dtop, dadd, ladd):
    """float64 forward + backward; returns activations, deltas, grads (list W, b per layer)."""
    xs = [torch.tensor(x, dtype=torch.float64)]
    params = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in arrays]
    hs = []
    cur = xs[0]
    for i, act in enumerate(acts):
        cur = cur @ params[2 * i] + params[2 * i + 1]
        if act == "tanh":
            cur = torch.tanh(cur)
        cur.retain_grad()
        hs.append(cur)
    obj = (hs[-1] * torch.tensor(dtop, dtype=torch.float64)).sum()
    if dadd is not None:
        obj = obj + (hs[ladd] * torch.tensor(dadd, dtype=torch.float64)).sum()
    obj.backward()
    deltas = []
    for i, act in enumerate(acts):
        dh = hs[i].grad
        deltas.append(dh * (1 - hs[i].detach() ** 2) if act == "tanh" else dh)
    return [h.detach().numpy() for h in hs], [d.numpy() for d in deltas], [p.grad.numpy() for p in params]


CASES = [
    # net_dim, n_neurons, activation_functions, B
    (128, [128, 128, 2], ["", "tanh", "tanh", ""], 4096),
    (20, [128, 128, 2], ["", "tanh", "tanh", ""], 256),
    (7, [64, 3], ["", "tanh", ""], 200),
    (1024, [128, 128, 2], ["", "tanh", "tanh", ""], 300),
    (33, [50, 17, 4], ["", "tanh", "tanh", "tanh"], 77),
    (6, [8, 2], ["", "tanh", ""], 5),
    (40, [32, 2], ["", "tanh", ""], 9000),     # more than 32 x 128 rows: the weight-gradient slices grow to 512 rows, ragged end
    (24, [40, 32, 24, 16, 3], ["", "tanh", "tanh", "tanh", "tanh", ""], 100),   # ten Dense layers (EMX_MAX_LAYERS = 12)
    (260, [300, 130, 2], ["", "tanh", "tanh", ""], 50),   # layers of several 128-column passes and 128-k chunks
    # source rows beyond the LDS images of the row-block kernels (~2400 columns): the range runs in segments, the wide layers through
    # the streaming kernel (csrc/wide.h) -- the 2600-wide ends of a 1300-feature periodic model ...
    (2600, [128, 128, 2], ["", "tanh", "tanh", ""], 70),
    # ... and wide layers INSIDE the network (every layer has a 2600-wide side; outputs of 21 passes: three pass groups per row block)
    (40, [2600, 3], ["", "tanh", ""], 50),
]


@pytest.mark.parametrize("net_dim,n_neurons,afs,B", CASES)
def test_mlp_forward_backward_wgrad(L, net_dim, n_neurons, afs, B):
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    rng = np.random.default_rng(net_dim * 31 + B)
    lay = build_layout(n_neurons, afs, net_dim)
    flat = init_parameters(lay, 2)
    arrays = unpack_weights(lay, flat)
    acts = [l.activation for l in lay.layers]
    n = len(lay.layers)
    x = rng.normal(0, 1, (B, net_dim)).astype(np.float32)
    dtop = rng.normal(0, 1, (B, net_dim)).astype(np.float32)
    ladd = lay.n_enc - 1
    dadd = rng.normal(0, 1, (B, lay.latent_dim)).astype(np.float32)
    hs, deltas, grads = reference(x, arrays, acts, dtop, dadd, ladd)

    dev = "cuda"
    params = torch.from_numpy(flat).to(dev)
    h = [torch.full((B, l.n_out), float("nan"), device=dev) for l in lay.layers]
    delta = [torch.full((B, l.n_out), float("nan"), device=dev) for l in lay.layers]
    plan = L.MlpPlan(lay, params, h, delta, train_rows=B)
    X = torch.from_numpy(x).to(dev)
    c0 = L.load().emx_launch_count()
    L.mlp_forward(plan, 0, n, X, 0, B)
    launches = L.load().emx_launch_count() - c0
    torch.cuda.synchronize()
    for i in range(n):
        np.testing.assert_allclose(h[i].cpu().numpy(), hs[i], rtol=2e-6, atol=3e-6, err_msg=f"activation {i}")
    # one launch per pass wherever the LDS images hold every layer's source; beyond ~2400 columns the range runs in segments: the
    # 2600-wide autoencoder as [streamed layer 0] [layers 1 .. 5 in one launch] (its wide OUTPUT keeps no image), the network with
    # 2600-wide layers inside layer by layer
    widest_source = max(l.n_in for l in lay.layers)
    assert launches == (1 if widest_source <= 2048 else (2 if net_dim == 2600 else n)), (launches, widest_source)
    # split ranges give the same numbers (encoder on all rows, decoder on a row block)
    h2 = [torch.full((B, l.n_out), float("nan"), device=dev) for l in lay.layers]
    plan2 = L.MlpPlan(lay, params, h2, delta)
    rb, re = B // 3, B - B // 5
    L.mlp_forward(plan2, 0, lay.n_enc, X, 0, B)
    L.mlp_forward(plan2, lay.n_enc, n, h2[lay.n_enc - 1], rb, re)
    torch.cuda.synchronize()
    assert torch.equal(h2[lay.n_enc - 1], h[lay.n_enc - 1])
    assert torch.equal(h2[-1][rb:re], h[-1][rb:re]) and torch.isnan(h2[-1][:rb]).all() and torch.isnan(h2[-1][re:]).all()

    DT, DA = torch.from_numpy(dtop).to(dev), torch.from_numpy(dadd).to(dev)
    L.mlp_backward(plan, 0, n, DT, DA, ladd, 0, B)
    torch.cuda.synchronize()
    for i in range(n):
        ref = deltas[i]
        np.testing.assert_allclose(delta[i].cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=f"delta {i}")
    # decoder and encoder back-propagated by separate launches (dh below the decoder handed over through `dbelow`)
    delta2 = [torch.full((B, l.n_out), float("nan"), device=dev) for l in lay.layers]
    plan3 = L.MlpPlan(lay, params, h, delta2)
    dlat = torch.full((B, lay.latent_dim), float("nan"), device=dev)
    L.mlp_backward(plan3, lay.n_enc, n, DT, None, 0, 0, B, dbelow=dlat)
    L.mlp_backward(plan3, 0, lay.n_enc, dlat, DA, ladd, 0, B)
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(delta2[i], delta[i]), f"split backward differs at layer {i}"
    g = torch.full((lay.n_params,), float("nan"), device=dev)
    ws = torch.zeros(L.mlp_wgrad_ws_floats(lay.n_params, B), device=dev)
    L.mlp_wgrad(plan, 0, n, 0, B, g, ws)
    torch.cuda.synchronize()
    gh = g.cpu().numpy()
    for i, l in enumerate(lay.layers):
        gw = gh[l.w_off: l.w_off + l.n_in * l.n_out].reshape(l.n_in, l.n_out)
        gb = gh[l.b_off: l.b_off + l.n_out]
        np.testing.assert_allclose(gw, grads[2 * i], rtol=0, atol=2e-5 * np.abs(grads[2 * i]).max(), err_msg=f"dW {i}")
        np.testing.assert_allclose(gb, grads[2 * i + 1], rtol=0, atol=2e-5 * np.abs(grads[2 * i + 1]).max(), err_msg=f"db {i}")
    # alignment padding between tensors stays exactly zero (the Adam kernel runs over the whole flat buffer)
    mask = np.ones(lay.n_params, bool)
    for l in lay.layers:
        mask[l.w_off: l.w_off + l.n_in * l.n_out] = False
        mask[l.b_off: l.b_off + l.n_out] = False
    lo = min(l.w_off for l in lay.layers)
    hi = max(l.b_off + l.n_out for l in lay.layers)
    pad = mask.copy()
    pad[:lo] = False
    pad[hi:] = False
    assert np.all(gh[pad] == 0)
    # deterministic
    g2 = torch.empty_like(g)
    L.mlp_wgrad(plan, 0, n, 0, B, g2, ws)
    torch.cuda.synchronize()
    assert torch.equal(g[lo:hi], g2[lo:hi])
    # the step's scalar epilogue riding along as one extra workgroup: same gradients, same sums and bookkeeping as the
    # separate emx_finish_step launch
    src = torch.from_numpy(rng.normal(0, 1, 5000).astype(np.float32)).to(dev)
    jobs = [(src, 5000, 0.5, None, 0, False), (src[1:], 777, 2.0, None, 2, True)]
    outs = []
    for fold in (False, True):
        terms = torch.tensor([9.0, 0.25, 9.0, 0.0], device=dev)
        hist, thist = torch.zeros(4, device=dev), torch.zeros(4, 4, device=dev)
        cnt = [torch.tensor([1], dtype=torch.int64, device=dev) for _ in range(3)]
        g3 = torch.empty_like(g)
        fin = L.make_finish_desc(jobs, terms, 4, hist, thist, *cnt)
        L.mlp_wgrad(plan, 0, n, 0, B, g3, ws, finish=fin if fold else None)
        if not fold:
            L.finish_step(jobs, terms, 4, hist, thist, *cnt)
        torch.cuda.synchronize()
        assert torch.equal(g3[lo:hi], g[lo:hi])
        assert [int(c) for c in cnt] == [2, 2, 2]
        outs.append((terms.clone(), hist.clone(), thist.clone()))
    for a, b in zip(*outs):     # (the embedded epilogue keeps fewer loads in flight: same sums, another association)
        torch.testing.assert_close(a, b, rtol=2e-6, atol=1e-7)
    assert float(outs[0][1][1]) == pytest.approx(float(outs[0][0].sum()), rel=1e-6) and float(outs[0][0][1]) == 0.25


def test_mlp_rejects_bad_arguments(L):
    from encodermap_amd.layout import build_layout, init_parameters

    lay = build_layout([8, 2], ["", "tanh", ""], 6)
    params = torch.from_numpy(init_parameters(lay, 0)).cuda()
    h = [torch.zeros(4, l.n_out, device="cuda") for l in lay.layers]
    plan = L.MlpPlan(lay, params, h, h, train_rows=4)
    x = torch.zeros(4, 6, device="cuda")
    with pytest.raises(L.EmxError, match="layer range"):
        L.mlp_forward(plan, 2, 1, x, 0, 4)
    with pytest.raises(L.EmxError, match="workspace"):
        L.mlp_wgrad(plan, 0, 4, 0, 4, torch.zeros(lay.n_params, device="cuda"), torch.zeros(8, device="cuda"))
    with pytest.raises(L.EmxError, match="multiple of 16"):     # the weight-gradient fragments are kept per 16-row block
        L.mlp_forward(plan, 0, 4, x, 1, 4)
    with pytest.raises(L.EmxError, match="operand store"):
        L.mlp_wgrad(L.MlpPlan(lay, params, h, h), 0, 4, 0, 4, torch.zeros(lay.n_params, device="cuda"), torch.zeros(8, device="cuda"))


def decode_operand_store(lay, mirror):
    """The weight operand store back to float64 kernels, per layer (forward operand, backward operand, both also with
    their zero padding), following the layout of csrc/wop.h: 64-byte headers (float 1/s per layer; region padded to 1 KB),
    then 64-KB units in execution order -- forward: layer, pass (128 columns of out), chunk (128 k of in); backward: layers
    descending, pass over in, chunk over out -- each [wave = tile of 16 columns][K-step of 32][plane] x 1 KB, lane
    (n = lane & 15, g = lane >> 4) holding B[32 ks + kk(g, e)][16 tile + n], kk(g, e) = 4 g + e (e < 4), 16 + 4 g + e - 4."""
    raw = mirror.cpu().numpy().view(np.uint8)
    L_ = len(lay.layers)
    units_begin = (L_ * 64 + 1023) // 1024 * 1024
    g128 = lambda x: (x + 127) // 128
    sizes = [g128(l.n_in) * g128(l.n_out) for l in lay.layers]
    fwd_unit = np.concatenate([[0], np.cumsum(sizes)])[:-1]
    bwd_unit = {}
    u = sum(sizes)
    for i in reversed(range(L_)):
        bwd_unit[i] = u
        u += sizes[i]
    assert units_begin + u * 65536 == raw.size
    lane = np.arange(64)
    n_, g_ = lane & 15, lane >> 4
    e = np.arange(8)
    kk = np.where(e < 4, 4 * g_[:, None] + e, 16 + 4 * g_[:, None] + e - 4)          # [lane, e]
    out = []
    for i, l in enumerate(lay.layers):
        K, N = l.n_in, l.n_out
        rs = float(raw[64 * i: 64 * i + 4].view(np.float32)[0])
        both = []
        for base_unit, kred, nout in ((fwd_unit[i], K, N), (bwd_unit[i], N, K)):
            P, C = g128(nout), g128(kred)
            full = np.zeros((128 * C, 128 * P))
            for p in range(P):
                for c in range(C):
                    o = units_begin + (base_unit + p * C + c) * 65536
                    unit = raw[o: o + 65536].view(np.float16).astype(np.float64).reshape(8, 4, 2, 64, 8)   # [wave, ks, plane, lane, e]
                    val = (unit[:, :, 0] + unit[:, :, 1]) * rs
                    for w in range(8):
                        for ks in range(4):
                            full[128 * c + 32 * ks + kk, (128 * p + 16 * w + n_)[:, None]] = val[w, ks]
            both.append(full)
        out.append((both[0][:K, :N], both[1][:N, :K].T, both[0], both[1]))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("net_dim,n_neurons", [(128, [128, 128, 2]), (37, [50, 3]), (300, [136, 20, 2])])
def test_adam_keeps_operand_store_current(L, net_dim, n_neurons):
    """The optimizer rewrites every kernel element at its two places in the operand store of the dense-layer kernels
    (include/emx.h, emx_mlp_mirror; layout csrc/wop.h): after an update both operands must decode to the new weights to
    22 bits (the scaled f16 pair), zero padded -- as a fresh emx_mlp_mirror does.  (The optimizer scales with the maximum
    recorded by the previous update, a fresh build with the exact one: the two stores may differ by a power of two in
    scale, not in the numbers they stand for.)"""
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    afs = [""] + ["tanh"] * (len(n_neurons) - 1) + [""]
    lay = build_layout(n_neurons, afs, net_dim)
    params = torch.from_numpy(init_parameters(lay, 3)).cuda()
    bufs = [torch.zeros(4, l.n_out, device="cuda") for l in lay.layers]
    plan = L.MlpPlan(lay, params, bufs, bufs)

    def check(mirror):
        ws = unpack_weights(lay, params.cpu().numpy())[0::2]
        for (fw, bw, fw_full, bw_full), W in zip(decode_operand_store(lay, mirror), ws):
            tol = 2.0 ** -21 * np.abs(W).max()
            np.testing.assert_allclose(fw, W, rtol=0, atol=tol)
            np.testing.assert_allclose(bw, W, rtol=0, atol=tol)
            K, N = W.shape
            assert not fw_full[K:].any() and not fw_full[:, N:].any() and not bw_full[N:].any() and not bw_full[:, K:].any()

    check(plan.mirror)
    g = torch.randn_like(params)
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
    before = plan.mirror.clone()
    for _ in range(2):
        L.adam_step(params, g, m, v, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, None, ws, plan=plan)
        step += 1
    assert not torch.equal(before, plan.mirror)
    check(plan.mirror)
    check(L.MlpPlan(lay, params, bufs, bufs).mirror)
    # slice-partial form
    parts = torch.randn(3, params.numel(), device="cuda")
    L.adam_step_slices(params, parts, 3, m, v, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, ws, plan=plan)
    check(plan.mirror)
    # the update itself is the one of the flat path, bit for bit (on the parameters: the flat path also walks the alignment
    # padding between tensors, whose gradient is zero in real use)
    real = torch.zeros(params.numel(), dtype=torch.bool)
    for l in lay.layers:
        real[l.w_off: l.w_off + l.n_in * l.n_out] = True
        real[l.b_off: l.b_off + l.n_out] = True
    real = real.cuda()
    p2, m2, v2 = params.clone(), m.clone(), v.clone()
    L.adam_step(params, g, m, v, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, None, ws, plan=plan)
    L.adam_step(p2, g, m2, v2, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, None, ws)
    assert torch.equal(params[real], p2[real]) and torch.equal(m[real], m2[real]) and torch.equal(v[real], v2[real])


@pytest.mark.gpu
@pytest.mark.parametrize("periodicity,variant,D,B", [(float("inf"), "mean_square", 128, 4096), (2 * np.pi, "mean_abs", 10, 256),
                                                     (360.0, "mean_norm", 37, 100), (float("inf"), "mean_norm", 70, 50),
                                                     (float("inf"), "mean_abs", 300, 40),
                                                     (2 * np.pi, "mean_abs", 512, 200),          # C4: 1024-wide ends
                                                     (float("inf"), "mean_square", 1100, 40)])
def test_forward_with_cost_equals_separate_kernels(L, periodicity, variant, D, B):
    """emx_mlp_forward_cost (dense layers + auto cost in one launch) against the two stand-alone entry points it
    replaces (emx_mlp_forward, emx_periodic_cost_fwd_bwd): same activations, same cost and the same gradient with
    respect to the output."""
    from encodermap_amd.layout import build_layout, init_parameters

    periodic = np.isfinite(periodicity)
    net_dim = 2 * D if periodic else D
    lay = build_layout([64, 48, 2], ["", "tanh", "tanh", ""], net_dim)
    n = len(lay.layers)
    params = torch.from_numpy(init_parameters(lay, 5)).cuda()
    g = torch.Generator().manual_seed(0)
    xb = ((torch.rand(B, D, generator=g) - 0.5) * (periodicity if periodic else 6.0)).cuda()
    xin = torch.zeros(B, net_dim, device="cuda")
    if periodic:
        L.periodic_in(xb, periodicity, xin)
    else:
        xin = xb
    coef = 0.7
    # separate kernels
    h1 = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
    dY1 = torch.zeros(B, net_dim, device="cuda")
    plan1 = L.MlpPlan(lay, params, h1, h1)
    L.mlp_forward(plan1, 0, n, xin, 0, B)
    loss1 = torch.zeros(1, device="cuda")
    ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
    L.periodic_cost_fwd_bwd(h1[-1], 0, D, periodic, xb, D, periodicity, variant, coef, 0, B, loss1, dY1, ws, B=B)
    # one launch
    h2 = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
    dY2 = torch.zeros(B, net_dim, device="cuda")
    plan2 = L.MlpPlan(lay, params, h2, h2)
    cost_ws = torch.zeros(L.mlp_row_blocks(B), device="cuda")
    cost = L.make_auto_cost(xb, D, 0, D, periodic, periodicity, variant, coef, B, dY2, cost_ws)
    L.mlp_forward(plan2, 0, n, xin, 0, B, cost=cost)
    torch.cuda.synchronize()
    for a, b in zip(h1, h2):
        assert torch.equal(a, b)
    np.testing.assert_allclose(dY2.cpu().numpy(), dY1.cpu().numpy(), rtol=1e-6, atol=1e-9)
    assert float(cost_ws.double().sum()) == pytest.approx(float(loss1), rel=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("periodicity,D,B", [(float("inf"), 128, 96), (2 * np.pi, 10, 96), (360.0, 33, 96),
                                             # rows whose width is a multiple of 4: two rows per wave on 16-byte lanes (odd B: a half-empty wave)
                                             (float("inf"), 36, 97), (2 * np.pi, 12, 97), (360.0, 200, 33), (float("inf"), 300, 5)])
def test_adam_gathers_the_next_batch(L, periodicity, D, B):
    """emx_adam_step(next) leaves the rows of batch cursor + 1 in xb / xin (and their split planes) exactly as emx_gather_rows
    would, skips the gather beyond the last batch of the index block, and still performs the same update."""
    N, nb = 500, 3
    periodic = np.isfinite(periodicity)
    g = torch.Generator().manual_seed(1)
    data = ((torch.rand(N, D, generator=g) - 0.5) * (periodicity if periodic else 4.0)).cuda()
    idx = torch.randint(0, N, (nb * B,), generator=g).cuda()
    n = 1000
    w = torch.randn(n, generator=g).cuda()
    grad = torch.randn(n, generator=g).cuda()
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
    for cur in range(nb):
        cursor = torch.tensor([cur], dtype=torch.int64, device="cuda")
        xb = torch.full((B, D), -7.0, device="cuda")
        xin = torch.full((B, 2 * D), -7.0, device="cuda") if periodic else None
        split = None if periodic else torch.full((L.sketchmap_split_bytes(B, D),), 0x5A, dtype=torch.uint8, device="cuda")
        nxt = L.make_next_batch(data, idx, cursor, B, periodicity, xb, xin, split)
        w1, m1, v1 = w.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        w2, m2, v2 = w.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        L.adam_step(w1, grad, m1, v1, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, None, ws, next_batch=nxt)
        L.adam_step(w2, grad, m2, v2, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, None, ws)
        assert torch.equal(w1, w2) and torch.equal(m1, m2) and torch.equal(v1, v2)
        if cur + 1 < nb:
            xb_ref = torch.zeros(B, D, device="cuda")
            xin_ref = torch.zeros(B, 2 * D, device="cuda") if periodic else None
            L.gather_rows(data, idx, periodicity, xb_ref, xin_ref, cursor=torch.tensor([cur + 1], device="cuda"), B=B)
            assert torch.equal(xb, xb_ref)
            assert torch.equal(xb, data[idx[(cur + 1) * B:(cur + 2) * B]])
            if periodic:
                assert torch.equal(xin, xin_ref)
            else:   # the split rows of the all-pairs kernel: planes bit for bit, norms to the order of their sums
                split_ref = torch.full_like(split, 0x5A)
                L.gather_rows(data, idx, periodicity, torch.zeros(B, D, device="cuda"), None, cursor=torch.tensor([cur + 1], device="cuda"), B=B,
                              split_ws=split_ref)
                torch.cuda.synchronize()
                n_plane = 2 * 2 * B * ((D + 31) // 32 * 32)
                assert torch.equal(split[:n_plane], split_ref[:n_plane])
                tail = split[n_plane:].view(torch.float32).cpu().numpy(), split_ref[n_plane:].view(torch.float32).cpu().numpy()
                np.testing.assert_allclose(tail[0][:B], tail[1][:B], rtol=2e-6)        # squared norms
                np.testing.assert_array_equal(tail[0][B:2 * B], tail[1][B:2 * B])      # 1 / scale
        else:
            assert float(xb.max()) == -7.0   # nothing gathered behind the last batch


@pytest.mark.gpu
@pytest.mark.parametrize("periodicity,variant,D,B,n_neurons", [
    (float("inf"), "mean_square", 128, 4096, [128, 128, 2]), (2 * np.pi, "mean_abs", 10, 256, [128, 128, 2]),
    (360.0, "mean_norm", 37, 100, [64, 48, 3]), (float("inf"), "mean_abs", 150, 40, [136, 20, 2])])
def test_forward_backward_in_one_launch_equals_two(L, periodicity, variant, D, B, n_neurons, monkeypatch):
    """emx_mlp_forward_backward (forward, auto cost, decoder backward in one launch, activations kept in LDS) against
    emx_mlp_forward_cost followed by emx_mlp_backward: the same activations, deltas, latent gradient and cost -- a statement about
    the row-block chain kernel (EMX_TCHAIN=0: the first case's network would otherwise take the transposed chain, whose own
    comparison with the separate launches -- to rounding, another order of summation -- is test_transposed_chain_against_the_oracle)."""
    from encodermap_amd.layout import build_layout, init_parameters

    monkeypatch.setenv("EMX_TCHAIN", "0")

    periodic = np.isfinite(periodicity)
    net_dim = 2 * D if periodic else D
    afs = [""] + ["tanh"] * (len(n_neurons) - 1) + [""]
    lay = build_layout(n_neurons, afs, net_dim)
    n = len(lay.layers)
    params = torch.from_numpy(init_parameters(lay, 9)).cuda()
    g = torch.Generator().manual_seed(2)
    xb = ((torch.rand(B, D, generator=g) - 0.5) * (periodicity if periodic else 6.0)).cuda()
    xin = torch.zeros(B, net_dim, device="cuda")
    if periodic:
        L.periodic_in(xb, periodicity, xin)
    else:
        xin = xb
    Ld = lay.layers[lay.n_enc - 1].n_out

    def run(fused):
        h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
        dl = [torch.full((B, l.n_out), float("nan"), device="cuda") for l in lay.layers]
        dlat = torch.full((B, Ld), float("nan"), device="cuda")
        plan = L.MlpPlan(lay, params, h, dl)
        cost_ws = torch.zeros(L.mlp_row_blocks(B), device="cuda")
        cost = L.make_auto_cost(xb, D, 0, D, periodic, periodicity, variant, 0.7, B, dl[-1], cost_ws)
        if fused:
            L.mlp_forward_backward(plan, 0, n, xin, 0, B, cost, lay.n_enc, dbelow=dlat)
        else:
            L.mlp_forward(plan, 0, n, xin, 0, B, cost=cost)
            L.mlp_backward(plan, lay.n_enc, n, dl[-1], None, 0, 0, B, dbelow=dlat)
        torch.cuda.synchronize()
        return h, dl, dlat, cost_ws

    h1, d1, z1, c1 = run(False)
    h2, d2, z2, c2 = run(True)
    for a, b in zip(h1, h2):
        assert torch.equal(a, b)
    assert torch.equal(c1, c2)
    for l in range(lay.n_enc, n):
        np.testing.assert_allclose(d2[l].cpu().numpy(), d1[l].cpu().numpy(), rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(z2.cpu().numpy(), z1.cpu().numpy(), rtol=1e-6, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("net_dim,n_neurons,B", [(40, [48, 3], 50), (130, [70, 33, 2], 37), (300, [160, 2], 16)])
def test_integer_data_is_exact(L, net_dim, n_neurons, B):
    """Lane maps of the matrix instructions, the transposed LDS images, the operand stores and the k order inside a K-step,
    checked with exact data (cdna_hip_programming.md section 3: 'check the map with exact integer data'): small integers
    are their own f16 hi planes, every product and sum below 2^24 is exact in float32, so a linear network must
    reproduce the integer matrix products bit for bit -- forward, backward and the weight gradients."""
    from encodermap_amd.layout import build_layout, pack_weights

    rng = np.random.default_rng(B)
    afs = [""] * (len(n_neurons) + 1)
    lay = build_layout(n_neurons, afs, net_dim)
    n = len(lay.layers)
    arrays = []
    for l in lay.layers:
        arrays += [rng.integers(-1, 2, (l.n_in, l.n_out)).astype(np.float32), rng.integers(-3, 4, l.n_out).astype(np.float32)]
    x = rng.integers(-4, 5, (B, net_dim)).astype(np.float32)
    dtop = rng.integers(-1, 2, (B, net_dim)).astype(np.float32)
    hs, cur = [], x.astype(np.int64)
    for i in range(n):
        cur = cur @ arrays[2 * i].astype(np.int64) + arrays[2 * i + 1].astype(np.int64)
        hs.append(cur)
    assert max(np.abs(h).max() for h in hs) < 2 ** 22      # 22 bits: exactly a scaled f16 pair
    deltas, d = [None] * n, dtop.astype(np.int64)
    for i in reversed(range(n)):
        deltas[i] = d
        d = d @ arrays[2 * i].astype(np.int64).T
    assert max(np.abs(d).max() for d in deltas) < 2 ** 22
    params = torch.from_numpy(pack_weights(lay, arrays)).cuda()
    h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
    dl = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
    plan = L.MlpPlan(lay, params, h, dl, train_rows=B)
    L.mlp_forward(plan, 0, n, torch.from_numpy(x).cuda(), 0, B)
    L.mlp_backward(plan, 0, n, torch.from_numpy(dtop).cuda(), None, 0, 0, B)
    g = torch.zeros(lay.n_params, device="cuda")
    ws = torch.zeros(L.mlp_wgrad_ws_floats(lay.n_params, B), device="cuda")
    L.mlp_wgrad(plan, 0, n, 0, B, g, ws)
    torch.cuda.synchronize()
    for i in range(n):
        assert np.array_equal(h[i].cpu().numpy().astype(np.int64), hs[i]), f"activation {i}"
        assert np.array_equal(dl[i].cpu().numpy().astype(np.int64), deltas[i]), f"delta {i}"
    gh = g.cpu().numpy().astype(np.int64)
    for i, l in enumerate(lay.layers):
        xin = x.astype(np.int64) if i == 0 else hs[i - 1]
        dW = xin.T @ deltas[i]
        assert np.abs(dW).max() < 2 ** 24
        assert np.array_equal(gh[l.w_off: l.w_off + l.n_in * l.n_out].reshape(l.n_in, l.n_out), dW), f"dW {i}"
        assert np.array_equal(gh[l.b_off: l.b_off + l.n_out], deltas[i].sum(0)), f"db {i}"


@pytest.mark.gpu
def test_wgrad_adam_local_without_next_batch_and_regulariser(L):
    """the optional arguments of emx_mlp_wgrad_adam_local left out (no next batch: no gather workgroups; no L2 coefficients): the launch
    still advances the counters, writes the history row and updates every parameter"""
    from encodermap_amd.layout import build_layout, init_parameters

    lay = build_layout([64, 2], ["", "tanh", ""], 40)
    B, n, dev = 300, len(lay.layers), "cuda"
    g = torch.Generator().manual_seed(2)
    params = torch.from_numpy(init_parameters(lay, 1)).to(dev)
    h = [torch.zeros(B, l.n_out, device=dev) for l in lay.layers]
    delta = [torch.zeros(B, l.n_out, device=dev) for l in lay.layers]
    plan = L.MlpPlan(lay, params, h, delta, train_rows=B)
    x = torch.randn(B, 40, generator=g).to(dev)
    L.mlp_forward(plan, 0, n, x, 0, B)
    L.mlp_backward(plan, 0, n, (torch.randn(B, 40, generator=g) * 0.01).to(dev), None, lay.n_enc - 1, 0, B)
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    step, cursor, hstep = (torch.tensor([k], dtype=torch.int64, device=dev) for k in (7, 3, 7))
    reg_ws = torch.zeros(2 * L.EMX_REDUCE_WS, device=dev)
    terms, hist, thist = torch.tensor([1.0, 2.0, 0.0, 0.0], device=dev), torch.zeros(16, device=dev), torch.zeros(16, 4, device=dev)
    fin = L.make_finish_desc([(reg_ws, L.EMX_REDUCE_WS, 1.0, None, 3, False)], terms, 4, hist, thist, hstep, cursor, step)
    sync = torch.zeros(4 + 1024, dtype=torch.int32, device=dev)
    w0 = params.clone()
    ws = torch.zeros(max(L.mlp_wgrad_adam_local_ws_floats(plan, B), 16), device=dev)
    L.mlp_wgrad_adam_local(plan, 0, B, ws, fin, m, v, None, 1e-3, 0.9, 0.999, 1e-7, 1.0, reg_ws, sync)
    torch.cuda.synchronize()
    assert [int(step), int(cursor), int(hstep)] == [8, 4, 8] and (sync.cpu().numpy() == 0).all()
    assert float(hist[7]) == 3.0 and float(reg_ws.abs().sum()) == 0.0          # (no L2 coefficients: no regulariser shares)
    changed = (params != w0).float().mean().item()
    assert changed > 0.95 and torch.isfinite(params).all()


@pytest.mark.gpu
@pytest.mark.parametrize("net_dim,n_neurons,afs,B", [CASES[i] for i in (0, 1, 2, 4, 5, 7, 8)] + [(128, [128, 128, 2], ["", "tanh", "tanh", ""], 1500)])
def test_wgrad_adam_local_through_the_abi(L, net_dim, n_neurons, afs, B):
    """emx_mlp_wgrad_adam_local called as a C consumer would (no engine): against emx_mlp_wgrad + emx_adam_step_slices(bookkeeping_done)
    on the same fragments -- new weights to 3e-3 of the update (Adam's first step is lr g / (|g| + eps): a rounding of g moves it),
    first moments to 5e-6 of their largest element, the operand store decodes to the new weights, the epilogue's sums and counters,
    the regulariser shares of the NEW weights in the half of ws the parity names, the next batch's rows bit for bit, sync left clean."""
    from encodermap_amd.layout import build_layout, init_parameters, l2_coefficients

    rng = np.random.default_rng(net_dim * 7 + B)
    lay = build_layout(n_neurons, afs, net_dim)
    flat = init_parameters(lay, 2)
    n = len(lay.layers)
    dev = "cuda"
    N = 3 * B
    data = torch.from_numpy(rng.normal(0, 1, (N, net_dim)).astype(np.float32)).to(dev)
    idx = torch.from_numpy(rng.integers(0, N, 3 * B)).to(dev)
    dtop = torch.from_numpy((rng.normal(0, 1, (B, net_dim)) * 0.05).astype(np.float32)).to(dev)
    l2c = torch.from_numpy(l2_coefficients(lay, 0.001)).to(dev)
    assert L.mlp_wgrad_adam_local_fits(L.MlpPlan(lay, torch.from_numpy(flat).to(dev), [torch.zeros(B, l.n_out, device=dev) for l in lay.layers],
                                                 [torch.zeros(B, l.n_out, device=dev) for l in lay.layers], train_rows=B), B)
    out = []
    for local in (False, True):
        params = torch.from_numpy(flat).to(dev)
        h = [torch.zeros(B, l.n_out, device=dev) for l in lay.layers]
        delta = [torch.zeros(B, l.n_out, device=dev) for l in lay.layers]
        plan = L.MlpPlan(lay, params, h, delta, train_rows=B)
        xb = data[idx[:B]].contiguous()
        L.mlp_forward(plan, 0, n, xb, 0, B)
        L.mlp_backward(plan, 0, n, dtop, None, lay.n_enc - 1, 0, B)
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        step = torch.tensor([4], dtype=torch.int64, device=dev)            # (an even number of updates so far: parity 0)
        cursor, hstep = torch.tensor([0], dtype=torch.int64, device=dev), torch.tensor([4], dtype=torch.int64, device=dev)
        reg_ws = torch.zeros(2 * L.EMX_REDUCE_WS, device=dev)
        L.l2_partials(params, l2c, reg_ws)
        L.l2_partials(params, l2c, reg_ws[L.EMX_REDUCE_WS:])
        terms = torch.zeros(8, device=dev)
        hist, thist = torch.zeros(16, device=dev), torch.zeros(16, 8, device=dev)
        jobs = [(reg_ws, L.EMX_REDUCE_WS, 1.0, None, 3, False), (dtop, dtop.numel(), 2.0, None, 1, True)]
        fin = L.make_finish_desc(jobs, terms, 8, hist, thist, hstep, cursor, step)
        xnext = torch.full((B, net_dim), -7.0, device=dev)
        nxt = L.make_next_batch(data, idx, cursor, B, float("inf"), xnext, None, None)
        ws = torch.zeros(max(L.mlp_wgrad_ws_floats(lay.n_params, B), L.mlp_wgrad_adam_local_ws_floats(plan, B)), device=dev)
        sync = torch.zeros(4 + 1024, dtype=torch.int32, device=dev)
        w0 = params.clone()
        if local:
            L.mlp_wgrad_adam_local(plan, 0, B, ws, fin, m, v, l2c, 1e-3, 0.9, 0.999, 1e-7, 1.0, reg_ws, sync, next_batch=nxt)
        else:
            L.mlp_wgrad(plan, 0, n, 0, B, None, ws, finish=fin)
            L.adam_step_slices(params, ws, L.mlp_wgrad_slices(B), m, v, l2c, 1e-3, 0.9, 0.999, 1e-7, 1.0, step, reg_ws, plan=plan, next_batch=nxt,
                               bookkeeping_done=True)
        torch.cuda.synchronize()
        assert (sync.cpu().numpy() == 0).all()
        half = reg_ws[L.EMX_REDUCE_WS:] if local else reg_ws[:L.EMX_REDUCE_WS]     # (update 5: parity 1; the two launches always use the first half)
        dec = decode_operand_store(lay, plan.mirror)
        out.append(dict(w=params.cpu().numpy(), w0=w0.cpu().numpy(), m=m.cpu().numpy(), v=v.cpu().numpy(), terms=terms.cpu().numpy(), hist=hist.cpu().numpy(),
                        counters=[int(step), int(cursor), int(hstep)], reg=float(half.double().sum()), xnext=xnext.cpu().numpy(), store=dec))
    a, b = out
    upd = np.abs(a["w"] - a["w0"]).max()
    dw = np.abs(a["w"] - b["w"])
    # (elements whose gradient all but cancels: there lr g / (|g| + eps) follows the last bits of g -- a handful, bounded separately)
    assert upd > 0 and np.quantile(dw, 0.999) <= 3e-3 * upd and dw.max() <= 0.1 * upd
    np.testing.assert_allclose(b["m"], a["m"], rtol=2e-5, atol=1e-9 + 5e-6 * np.abs(a["m"]).max())
    np.testing.assert_allclose(b["v"], a["v"], rtol=1e-4, atol=1e-12 + 1e-5 * np.abs(a["v"]).max())
    np.testing.assert_array_equal(a["terms"], b["terms"])
    np.testing.assert_array_equal(a["hist"], b["hist"])
    assert a["counters"] == b["counters"] == [5, 1, 5]
    np.testing.assert_array_equal(a["xnext"], b["xnext"])
    assert (a["xnext"] != -7.0).all()
    assert b["reg"] == pytest.approx(a["reg"], rel=1e-5) and a["reg"] > 0
    # the operand store holds the NEW weights (22-bit pairs: 2^-21 of the layer's largest weight)
    from encodermap_amd.layout import unpack_weights
    wn = unpack_weights(lay, b["w"])
    for l in range(n):
        np.testing.assert_allclose(b["store"][l][0], wn[2 * l], rtol=0, atol=2.0 ** -20 * np.abs(wn[2 * l]).max())


@pytest.mark.gpu
def test_transposed_chain_against_the_oracle(L):
    """The fused dense-layer launch as the TRANSPOSED, register-resident chain (csrc/tchain.h; the default up to one row block per compute unit,
    EMX_TCHAIN=1: at any size -- a process of its own keeps the switch away from the other tests): activations, the output rows, decoder deltas, latent gradient and cost against the separate
    row-block launches and float64, and the weight gradients formed from the fragments it leaves (tests/tchain_check.py: full and
    ragged row blocks, narrow inputs, a 3-wide latent)."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "tchain_check.py")], env=dict(os.environ, EMX_TCHAIN="1"), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
