"""Reads the in-kernel timestamps of an ABL_TIMING build of mlp_forward_kernel (kernel experiments)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from encodermap_amd.layout import build_layout, init_parameters
lib = _lib.load()
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096   # rows
D = int(sys.argv[4]) if len(sys.argv) > 4 else 128    # width of the network's input (2 x features for periodic data)
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 0)).cuda()
h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
dl = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
plan = _lib.MlpPlan(lay, params, h, dl)
x = torch.randn(B, D, device="cuda")
fusedbwd = len(sys.argv) > 2 and sys.argv[2] == "fusedbwd"
if fusedbwd:
    xb = x; dlat = torch.zeros(B, 2, device="cuda"); cws = torch.zeros(_lib.mlp_row_blocks(B), device="cuda")
    cost = _lib.make_auto_cost(xb, D, 0, D, False, float("inf"), "mean_square", 1.0, B, dl[-1], cws)
    for _ in range(20): _lib.mlp_forward_backward(plan, 0, n, x, 0, B, cost, lay.n_enc, dbelow=dlat)
    n = n + 3   # units: 6 forward + 3 backward
else:
    for _ in range(20): _lib.mlp_forward(plan, 0, n, x, 0, B)
torch.cuda.synchronize()
buf = np.zeros(8192, dtype=np.int64)
lib.emx_debug_read.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read(buf.ctypes.data) == 0
for blk in ((0, 1) if B >= 16 * 101 else (0,)):   # (the kernel records workgroups 0 and 100)
    clk = buf[blk * 2048: blk * 2048 + 512].reshape(8, 64); wall = clk * 0
    nslot = 3 + 5 * n
    for w in (0, 3, 7):
        c = clk[w, :nslot] - clk[w, 0]; t = c * 0
        print(f"block {blk*100} wave {w}: last unit's barrier passed at {c[-1]} clk after kernel entry")
        cc = clk[w] - clk[w, 0]
        print("  clk  stage loads issued:", c[1], " staged + barrier:", c[2], " loop end:", cc[60], " cost phase end:", cc[61])
        for l in range(n):
            s3, s5, s6, s7 = (c[3 + 5 * l], c[5 + 5 * l], c[6 + 5 * l], c[7 + 5 * l])
            prev = c[2] if l == 0 else c[7 + 5 * (l - 1)]
            print(f"   u{l}: start {prev:6d}  head {s3-prev:5d}  mfma+requests {s5-s3:5d}  epilogue {s6-s5:5d}  barrier {s7-s6:5d}")
