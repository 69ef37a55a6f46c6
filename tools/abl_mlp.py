"""Times mlp_forward/backward for growing layer ranges with an alternative build of the library (kernel experiments)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
plan = _lib.MlpPlan(lay, params, h, dl)
x = torch.randn(B, D, device="cuda"); dY = torch.randn(B, D, device="cuda")
def timeit(fn, reps=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = [os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "default"]
out.append("fwd[0,k) k=1..6: " + " ".join(f"{timeit(lambda: _lib.mlp_forward(plan, 0, k, x, 0, B)):6.2f}" for k in range(1, n + 1)))
out.append("bwd[k,6) k=5..0: " + " ".join(f"{timeit(lambda: _lib.mlp_backward(plan, k, n, dY, None, 0, 0, B)):6.2f}" for k in range(n - 1, -1, -1)))
print(" | ".join(out))
