"""A few launches of one dense-layer kernel at the C2 shape, for counter collection (tools/run_pmc_chain.sh): fused | fwd | bwd | wgrad"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
from encodermap_amd.layout import build_layout, init_parameters
_lib.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
B, D = 4096, 128
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 0)).cuda()
h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
dl = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
x = torch.randn(B, D, device="cuda")
plan = _lib.MlpPlan(lay, params, h, dl, train_rows=B)
dlat = torch.zeros(B, 2, device="cuda"); dZ = torch.randn(B, 2, device="cuda"); cws = torch.zeros(_lib.mlp_row_blocks(B), device="cuda")
cost = _lib.make_auto_cost(x, D, 0, D, False, float("inf"), "mean_abs", 1.0, B, dl[-1], cws)
g = torch.zeros(lay.n_params, device="cuda"); ws = torch.zeros(_lib.mlp_wgrad_ws_floats(lay.n_params, B), device="cuda")
_lib.mlp_forward_backward(plan, 0, n, x, 0, B, cost, lay.n_enc, dbelow=dlat)
_lib.mlp_backward(plan, 0, lay.n_enc, dlat, dZ, lay.n_enc - 1, 0, B)
for _ in range(30):
    if mode == "fused": _lib.mlp_forward_backward(plan, 0, n, x, 0, B, cost, lay.n_enc, dbelow=dlat)
    elif mode == "fwd": _lib.mlp_forward(plan, 0, n, x, 0, B)
    elif mode == "bwd": _lib.mlp_backward(plan, 0, lay.n_enc, dlat, dZ, lay.n_enc - 1, 0, B)
    else: _lib.mlp_wgrad(plan, 0, n, 0, B, None, ws)
torch.cuda.synchronize()
