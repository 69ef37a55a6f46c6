"""Times the fused MLP kernels alone (C2 shapes) with HIP events; run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
from encodermap_amd.layout import build_layout, init_parameters
_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 0)).cuda()
h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
dl = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
plan = _lib.MlpPlan(lay, params, h, dl, train_rows=B)
cost_ws = torch.zeros(_lib.mlp_row_blocks(B), device="cuda")
dlat = torch.zeros(B, 2, device="cuda")
x = torch.randn(B, D, device="cuda"); dY = torch.randn(B, D, device="cuda"); dZ = torch.randn(B, 2, device="cuda")
g = torch.zeros(lay.n_params, device="cuda"); ws = torch.zeros(_lib.mlp_wgrad_ws_floats(lay.n_params, B), device="cuda")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("B", B, "D", D)
print("forward  us", timeit(lambda: _lib.mlp_forward(plan, 0, n, x, 0, B)))
print("backward us", timeit(lambda: _lib.mlp_backward(plan, 0, n, dY, dZ, lay.n_enc - 1, 0, B)))
print("wgrad (+ reduce) us", timeit(lambda: _lib.mlp_wgrad(plan, 0, n, 0, B, g, ws)))
print("wgrad (slices only) us", timeit(lambda: _lib.mlp_wgrad(plan, 0, n, 0, B, None, ws)))
cost = _lib.make_auto_cost(x, D, 0, D, False, float("inf"), "mean_abs", 1.0, B, dl[-1], cost_ws)
print("forward + cost + decoder backward us", timeit(lambda: _lib.mlp_forward_backward(plan, 0, n, x, 0, B, cost, lay.n_enc, dbelow=dlat)))
print("encoder backward us", timeit(lambda: _lib.mlp_backward(plan, 0, lay.n_enc, dlat, dZ, lay.n_enc - 1, 0, B)))
print("forward (inference, no fragments) us", timeit(lambda: _lib.mlp_forward(plan, 0, n, x, 0, B, train=False)))
for l in range(n):
    src = x if l == 0 else h[l - 1]
    print(f"forward layer {l} ({lay.layers[l].n_in}->{lay.layers[l].n_out}) us", timeit(lambda: _lib.mlp_forward(plan, l, l + 1, src, 0, B, train=False)))
for l in range(n):
    print(f"wgrad layer {l} (slices only) us", timeit(lambda: _lib.mlp_wgrad(plan, l, l + 1, 0, B, None, ws)))
for rows in (128, 512, 1024, 2048):
    print(f"wgrad rows [0, {rows}) (slices only) us", timeit(lambda: _lib.mlp_wgrad(plan, 0, n, 0, rows, None, ws)))
