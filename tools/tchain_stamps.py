"""Where a row block of the transposed chain (csrc/tchain.h) spends its time: s_memtime stamps of workgroup 0, every wave, per step.
Needs a build with -DTC_STAMP (tools/build_variant.sh - build_ab/libemx_tc_stamp.so "-DTC_STAMP"), EMX_LIB_PATH pointing at it and
EMX_TCHAIN=1.  usage: python tools/tchain_stamps.py [B]      run on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from encodermap_amd import _lib as L
from encodermap_amd.layout import build_layout, init_parameters

assert os.environ.get("EMX_TCHAIN") == "1"
lib = L.load()
D, n_neurons, B = 128, [128, 128, 2], int(sys.argv[1]) if len(sys.argv) > 1 else 256
afs = [""] + ["tanh"] * (len(n_neurons) - 1) + [""]
lay = build_layout(n_neurons, afs, D)
n = len(lay.layers)
params = torch.from_numpy(init_parameters(lay, 9)).cuda()
xb = ((torch.rand(B, D) - 0.5) * 6.0).cuda()
h = [torch.zeros(B, l.n_out, device="cuda") if i < n - 1 else None for i, l in enumerate(lay.layers)]
dl = [torch.zeros(B, l.n_out, device="cuda") for l in lay.layers]
dlat = torch.zeros(B, lay.layers[lay.n_enc - 1].n_out, device="cuda")
plan = L.MlpPlan(lay, params, h, dl, train_rows=B)
cost_ws = torch.zeros(L.mlp_row_blocks(B), device="cuda")
cost = L.make_auto_cost(xb, D, 0, D, False, float("inf"), "mean_abs", 0.7, B, dl[-1], cost_ws)
for _ in range(5):
    L.mlp_forward_backward(plan, 0, n, xb, 0, B, cost, lay.n_enc, dbelow=dlat)
torch.cuda.synchronize()
st = np.zeros((8, 32, 8), dtype=np.uint64)
lib.emx_debug_tc_stamps.argtypes = [C.c_void_p]
assert lib.emx_debug_tc_stamps(st.ctypes.data) == 0
st = st.astype(np.int64)
t0 = st[:, 30, 0].min()
names = ["top", "products+share", "barrier", "sums", "epilogue", "operand", "fragments"]
print("s_memtime ticks (shader-clock cycles on this part: ~54 000 of them were a 22-us launch) relative to the first wave's entry; min..max over the eight waves")
print("prologue: entry", st[:, 30, 0] - t0, "\n tables stored", st[:, 30, 3] - t0, "\n input masked", st[:, 30, 4] - t0, "\n operand", st[:, 30, 5] - t0)
nsteps = 2 * n - lay.n_enc
for s in range(nsteps):
    row = st[:, s, :7] - t0
    print(f"step {s}: " + "  ".join(f"{names[k]} {row[:, k].min()}..{row[:, k].max()}" for k in range(7)))
print("end", st[:, 31, 0] - t0)
