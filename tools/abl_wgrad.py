"""Timeline of the weight-gradient kernel (EMX_CHAIN_TIMING build): stamps of workgroup 0 and start/end of every workgroup."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from encodermap_amd.layout import build_layout, init_parameters
lib = _lib.load()
B, D = 4096, 128
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 0)).cuda()
h = [torch.randn(B, l.n_out, device="cuda") for l in lay.layers]
dl = [torch.randn(B, l.n_out, device="cuda") for l in lay.layers]
plan = _lib.MlpPlan(lay, params, h, dl)
x = torch.randn(B, D, device="cuda")
ws = torch.zeros(_lib.mlp_wgrad_ws_floats(lay.n_params, B), device="cuda")
for _ in range(10): _lib.mlp_wgrad(plan, 0, n, x, 0, B, None, ws)
torch.cuda.synchronize()
buf = np.zeros(3 * 1024 + 64, dtype=np.int64)
lib.emx_debug_read_wg.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read_wg(buf.ctypes.data) == 0
c = buf[:10] - buf[0]
print("workgroup 0 (clk): first loads issued", c[1], " first two batches done", c[2], " row loop done", c[3], " end", c[9])
nb = _lib.mlp_wgrad_slices(B) * sum(((l.n_in + 63) // 64) * ((l.n_out + 63) // 64) for l in lay.layers)
st = buf[64:64 + 3 * nb:3].astype(np.float64); en = buf[65:65 + 3 * nb:3].astype(np.float64)
t0 = st.min(); st = (st - t0) * 1e-2; en = (en - t0) * 1e-2
print("workgroups", nb, " span us", en.max().round(2), " starts pct 50/90/100", np.percentile(st, [50, 90, 100]).round(2),
      " lifetimes pct 10/50/90/100", np.percentile(en - st, [10, 50, 90, 100]).round(2))
