"""per-prefix check of the streaming inference launch on one network: which layer range first disagrees with float64 (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from encodermap_amd import _lib as L
from encodermap_amd.layout import build_layout, init_parameters, unpack_weights
from test_infer_gpu import reference

D, n_neurons, afs, rows = 33, [50, 17, 4], ["", "relu", "sigmoid", "tanh"], 200
rng = np.random.default_rng(1)
lay = build_layout(n_neurons, afs, D)
flat = init_parameters(lay, 3)
arrays = unpack_weights(lay, flat)
acts = [l.activation for l in lay.layers]
params = torch.from_numpy(flat).cuda()
n = len(lay.layers)
plan = L.MlpPlan(lay, params, [None] * n, [None] * n)
x = rng.normal(0, 1, (rows, D)).astype(np.float32)
X = torch.from_numpy(x).cuda()
for l0 in range(0, 3):
    for l1 in range(l0 + 1, 4):
        xin = x if l0 == 0 else reference(x, arrays, acts, 0, l0, None, None).astype(np.float32)
        out = torch.empty(rows, lay.layers[l1 - 1].n_out, device="cuda")
        L.mlp_infer(plan, l0, l1, torch.from_numpy(np.ascontiguousarray(xin)).cuda(), out)
        torch.cuda.synchronize()
        want = reference(xin, arrays, acts, l0, l1, None, None)
        print(f"layers [{l0}, {l1}) acts {acts[l0:l1]} widths {[lay.layers[i].n_in for i in range(l0, l1)]} -> {lay.layers[l1-1].n_out}: max abs err {np.abs(out.cpu().numpy() - want).max():.3e}", flush=True)
