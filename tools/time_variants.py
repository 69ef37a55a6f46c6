"""Times the C2 dense-layer launches with an alternative build of the library: python tools/time_variants.py <libemx.so> [env...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from encodermap_amd.layout import build_layout, init_parameters
_lib.load()
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
ws = torch.zeros(_lib.mlp_wgrad_ws_floats(lay.n_params, B), device="cuda")
def timeit(fn, reps=100):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(os.path.basename(_lib.LIB_PATH), {k: v for k, v in os.environ.items() if k.startswith("EMX_")})
print("  fused     %.2f us" % timeit(lambda: _lib.mlp_forward_backward(plan, 0, n, x, 0, B, cost, lay.n_enc, dbelow=dlat)))
print("  forward   %.2f us" % timeit(lambda: _lib.mlp_forward(plan, 0, n, x, 0, B)))
print("  enc. bwd  %.2f us" % timeit(lambda: _lib.mlp_backward(plan, 0, lay.n_enc, dlat, dZ, lay.n_enc - 1, 0, B)))
print("  wgrad     %.2f us" % timeit(lambda: _lib.mlp_wgrad(plan, 0, n, 0, B, None, ws)))
