"""Reads the in-kernel timestamps of an EMX_CHAIN_TIMING build of the row-block chain kernel (kernel experiments).
   hipcc (the build line of __graft_entry__.build, with -mllvm -amdgpu-kernarg-preload-count=16) -DEMX_CHAIN_TIMING -o build_ab/libemx_timing.so ; python tools/abl_timing.py build_ab/libemx_timing.so [fused|fwd] [B] [D]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from encodermap_amd.layout import build_layout, init_parameters
lib = _lib.load()
mode = sys.argv[2] if len(sys.argv) > 2 else "fwd"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096   # rows
D = int(sys.argv[4]) if len(sys.argv) > 4 else 128    # width of the network's input (2 x features for periodic data)
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 0)).cuda()
h = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
dl = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
plan = _lib.MlpPlan(lay, params, h, dl, train_rows=B)
x = torch.randn(B, D, device="cuda")
if mode == "fused":
    dlat = torch.zeros(B, 2, device="cuda"); cws = torch.zeros(_lib.mlp_row_blocks(B), device="cuda")
    cost = _lib.make_auto_cost(x, D, 0, D, False, float("inf"), "mean_square", 1.0, B, dl[-1], cws)
    for _ in range(20): _lib.mlp_forward_backward(plan, 0, n, x, 0, B, cost, lay.n_enc, dbelow=dlat)
    n = n + 3   # units: 6 forward + 3 backward
elif mode == "bwd_latent":   # the same with the latent gradient joined from the all-pairs kernel's per-tile partial sums (unchanged between launches)
    dlat = torch.randn(B, 2, device="cuda")
    _lib.mlp_forward(plan, 0, n, x, 0, B)
    Z = h[lay.n_enc - 1]
    sig = (16.0, 12, 6, 1, 2, 6)
    ws = torch.zeros(_lib.sketchmap_ws_bytes(B, 2) // 4, dtype=torch.float32, device="cuda")
    _lib.sketchmap_tiles(x, Z, float("inf"), sig, 0, B, ws, None)
    sk = _lib.sketchmap_describe(x, Z, float("inf"), sig, 500.0, 1e-4, 0, B, ws)
    for _ in range(20): _lib.mlp_backward(plan, 0, lay.n_enc, dlat, None, lay.n_enc - 1, 0, B, latent=sk.grad)
    n = lay.n_enc
elif mode == "bwd":     # the encoder's backward launch of the step (latent gradient joined in)
    dlat = torch.randn(B, 2, device="cuda"); dZ = torch.randn(B, 2, device="cuda")
    _lib.mlp_forward(plan, 0, n, x, 0, B)
    for _ in range(20): _lib.mlp_backward(plan, 0, lay.n_enc, dlat, dZ, lay.n_enc - 1, 0, B)
    n = lay.n_enc
else:
    for _ in range(20): _lib.mlp_forward(plan, 0, n, x, 0, B)
torch.cuda.synchronize()
buf = np.zeros(8192, dtype=np.int64)
lib.emx_debug_read.argtypes = [ctypes.c_void_p]
assert lib.emx_debug_read(buf.ctypes.data) == 0
for blk in ((0, 1) if B >= 16 * 101 else (0,)):   # (the kernel records workgroups 0 and 100)
    clk = buf[blk * 2048: blk * 2048 + 512].reshape(8, 64)
    for w in (0, 7):
        c = clk[w] - clk[w, 0]
        print(f"block {blk*100} wave {w}:  stage loads issued {c[1]}  staged + barrier {c[2]}  loop end {c[60]}  end {c[61]}")
        if mode == "fused":
            print(f"   fused: forward units done {c[49]}  packages requested, cost starts {c[50]}  cost done {c[51]}  barrier {c[52]}  re-staged {c[53]}  barrier {c[54]}")
        for l in range(n):
            s3, s4, s5, s6, s7 = (c[3 + 5 * l + i] for i in range(5))
            prev = c[2] if l == 0 else c[7 + 5 * (l - 1)]
            print(f"   u{l}: start {prev:6d}  head {s3-prev:5d}  mfma+requests {s4-s3:5d}  records {s5-s4:5d}  epilogue {s6-s5:5d}  barrier {s7-s6:5d}")
