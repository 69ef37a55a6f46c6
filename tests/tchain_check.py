"""Run by tests/test_mlp_gpu.py::test_transposed_chain_against_the_oracle in a process of its own with EMX_TCHAIN=1 (the kernel at
any number of row blocks, not only where it is the default): the fused dense-layer launch as the transposed, register-resident chain (csrc/tchain.h) against the separate
row-block launches and the float64 reference -- activations, deltas, latent gradient, cost, and the weight gradients formed from the
fragments it leaves.  Prints one line per case; exits non-zero on a mismatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from encodermap_amd import _lib as L
from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

assert os.environ.get("EMX_TCHAIN") == "1"
L.require_gpu()
L.load()
# D, n_neurons, rows of the buffers (= the batch size of the cost's normalisation), cost, first row of the launch (a data-parallel
# rank's run of row blocks starts inside the batch)
CASES = [(128, [128, 128, 2], 4096, "mean_abs", 0), (128, [128, 128, 2], 300, "mean_square", 0), (50, [64, 3], 77, "mean_abs", 0),
         (20, [128, 96, 2], 16, "mean_square", 0), (128, [128, 128, 2], 1000, "mean_abs", 208), (33, [96, 2], 160, "mean_square", 48)]
for D, n_neurons, B, variant, rb in CASES:
    afs = [""] + ["tanh"] * (len(n_neurons) - 1) + [""]
    lay = build_layout(n_neurons, afs, D)
    n = len(lay.layers)
    flat = init_parameters(lay, 9)
    params = torch.from_numpy(flat).cuda()
    g = torch.Generator().manual_seed(2)
    xb = ((torch.rand(B, D, generator=g) - 0.5) * 6.0).cuda()
    Ld = lay.layers[lay.n_enc - 1].n_out

    def run(fused):
        h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
        dl = [torch.full((B, l.n_out), float("nan"), device="cuda") for l in lay.layers]
        dlat = torch.full((B, Ld), float("nan"), device="cuda")
        plan = L.MlpPlan(lay, params, h, dl, train_rows=B)
        cost_ws = torch.zeros(L.mlp_row_blocks(B), device="cuda")
        cost = L.make_auto_cost(xb, D, 0, D, False, float("inf"), variant, 0.7, B, dl[-1], cost_ws)
        if fused:
            L.mlp_forward_backward(plan, 0, n, xb, rb, B, cost, lay.n_enc, dbelow=dlat)
        else:
            L.mlp_forward(plan, 0, n, xb, rb, B, cost=cost)
            L.mlp_backward(plan, lay.n_enc, n, dl[-1], None, 0, rb, B, dbelow=dlat)
        L.mlp_backward(plan, 0, lay.n_enc, dlat, None, 0, rb, B)         # (encoder: the row-block launch in both runs)
        gw = torch.full((lay.n_params,), float("nan"), device="cuda")
        ws = torch.zeros(L.mlp_wgrad_ws_floats(lay.n_params, B), device="cuda")
        L.mlp_wgrad(plan, 0, n, rb, B, gw, ws)
        torch.cuda.synchronize()
        return h, dl, dlat, cost_ws, gw

    h1, d1, z1, c1, g1 = run(False)
    h2, d2, z2, c2, g2 = run(True)
    # float64 reference of the same pass
    arrays = [torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in unpack_weights(lay, flat)]
    cur = xb.cpu().double()[rb:]
    hs = []
    for i, l in enumerate(lay.layers):
        cur = cur @ arrays[2 * i] + arrays[2 * i + 1]
        if l.activation == "tanh":
            cur = torch.tanh(cur)
        hs.append(cur)
    diff = hs[-1] - xb.cpu().double()[rb:]
    cost = 0.7 * (diff.abs().sum() if variant == "mean_abs" else (diff * diff).sum()) / (B * D)      # (normalised by the whole batch)
    grads = torch.autograd.grad(cost, arrays)
    different = False
    for i in range(n):
        a, b, r = h1[i].cpu().numpy()[rb:], h2[i].cpu().numpy()[rb:], hs[i].detach().numpy()
        assert not h2[i][:rb].any(), "rows in front of the launch's range were written"
        np.testing.assert_allclose(b, r, rtol=2e-6, atol=3e-6, err_msg=f"activation {i}")
        different |= not np.array_equal(a, b)
    assert different, "the fused launch gave the row-block kernel's bits: the transposed chain did not run"
    np.testing.assert_allclose(float(c2.sum()), float(cost.detach()), rtol=2e-6)
    np.testing.assert_allclose(float(c2.sum()), float(c1.sum()), rtol=2e-6)
    for l in range(lay.n_enc, n):
        ref = d1[l].cpu().numpy()[rb:]
        np.testing.assert_allclose(d2[l].cpu().numpy()[rb:], ref, rtol=0, atol=2e-5 * np.abs(ref).max(), err_msg=f"delta {l}")
    np.testing.assert_allclose(z2.cpu().numpy()[rb:], z1.cpu().numpy()[rb:], rtol=0, atol=2e-5 * np.abs(z1.cpu().numpy()[rb:]).max())
    gh = g2.cpu().numpy()
    for i, l in enumerate(lay.layers):
        gw = gh[l.w_off: l.w_off + l.n_in * l.n_out].reshape(l.n_in, l.n_out)
        gb = gh[l.b_off: l.b_off + l.n_out]
        np.testing.assert_allclose(gw, grads[2 * i].numpy(), rtol=0, atol=2e-5 * np.abs(grads[2 * i].numpy()).max(), err_msg=f"dW {i}")
        np.testing.assert_allclose(gb, grads[2 * i + 1].numpy(), rtol=0, atol=2e-5 * np.abs(grads[2 * i + 1].numpy()).max(), err_msg=f"db {i}")
    print(f"transposed chain ok: D {D} n_neurons {n_neurons} rows {rb}..{B} {variant}", flush=True)
print("ALL OK")
