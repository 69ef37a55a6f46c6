"""Where a wave of the streaming inference launch (csrc/infer.hip) spends a 32-row block: s_memtime stamps of workgroup 0's waves.
Needs the stamp build (tools/build_infer_variant.sh build_ab/libemx_in_stamp.so "-DIN_STAMP") and EMX_LIB_PATH pointing at it.
usage: python tools/infer_stamps.py [encode|decode] [rows]      run on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from encodermap_amd import _lib as L
from encodermap_amd.layout import build_layout, init_parameters

which = sys.argv[1] if len(sys.argv) > 1 else "encode"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
lib = L.load()
lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], 128)
n, ne = len(lay.layers), lay.n_enc
params = torch.from_numpy(init_parameters(lay, 9)).cuda()
plan = L.MlpPlan(lay, params, [None] * n, [None] * n)
x = torch.randn(rows, 128 if which == "encode" else 2, device="cuda")
out = torch.empty(rows, 2 if which == "encode" else 128, device="cuda")
l0, l1 = (0, ne) if which == "encode" else (ne, n)
for _ in range(3):
    L.mlp_infer(plan, l0, l1, x, out)
torch.cuda.synchronize()
st = np.zeros((8, 16, 8), dtype=np.uint64)
lib.emx_debug_infer_stamps.argtypes = [C.c_void_p]
assert lib.emx_debug_infer_stamps(st.ctypes.data) == 0
st = st.astype(np.int64)
waves = 8
t0 = st[:waves, 15, 0].min()
print(f"{which}, {rows} rows; s_memtime ticks relative to the first wave's entry, per wave")
print("pieces built after", st[:waves, 15, 1] - t0)
names = ["top", "operand", "layer0", "layer1", "layer2", "layer3", "last", "stored"]
for it in range(8):
    for w in range(waves):
        row = st[w, it]
        if row[0] == 0:
            continue
        d = {names[k]: int(row[k] - row[0]) for k in range(8) if row[k]}
        print(f"iteration {it} wave {w}: top at {int(row[0] - t0)}  " + "  ".join(f"{k} +{v}" for k, v in d.items() if k != "top"))
