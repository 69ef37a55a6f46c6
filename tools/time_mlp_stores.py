import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from encodermap_amd import _lib
from encodermap_amd.layout import build_layout, init_parameters
_lib.load()
B, D = 4096, 128
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 0)).cuda()
h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
x = torch.randn(B, D, device="cuda")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, hh, tr in (("all f32 rows + fragments", h, B), ("all f32 rows, no fragments", h, 0), ("no f32 rows, fragments", [None] * (n - 1) + [h[-1]], B),
                     ("last layer's rows only", [None] * (n - 1) + [h[-1]], 0)):
    plan = _lib.MlpPlan(lay, params, hh, [None] * n, train_rows=tr)
    print(f"{name:32s} forward {timeit(lambda: _lib.mlp_forward(plan, 0, n, x, 0, B)):.2f} us")
