"""Slot timeline of the phase-alternating all-pairs kernel: s_memtime stamps of workgroup 0, wave 0 of each group (build with
tools/build_sk_variant.sh build_ab/libemx_pp_stamp.so "-DPP_STAMP", EMX_LIB_PATH).  Run on the GPU box."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib as L

lib = L.load()
B, D = 16384, 256
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(B, D, generator=g, device="cuda") * 2
z = torch.randn(B, 2, generator=g, device="cuda")
sig = (22.6, 12, 6, 1, 2, 6)
split = torch.zeros(L.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
L.sketchmap_split(x, split)
ws = torch.zeros(L.sketchmap_ws_bytes(B, 2) // 4, dtype=torch.float32, device="cuda")
for _ in range(3):
    L.sketchmap_tiles(x, z, float("inf"), sig, 0, B, ws, split)
torch.cuda.synchronize()
st = np.zeros((2, 64, 8), dtype=np.uint64)
lib.emx_debug_pp_stamps.argtypes = [C.c_void_p]
assert lib.emx_debug_pp_stamps(st.ctypes.data) == 0
st = st.astype(np.int64)
t0 = st[0, 0, 0]
print("slot: group A (top, +requests, +work, +waits, +barrier) | group B ...   [ticks]")
for s in range(2, 50):
    row = []
    for gq in range(2):
        r = st[gq, s]
        row.append(f"{int(r[0]-t0):8d} +{int(r[1]-r[0]) if r[1] else 0:5d} +{int(r[2]-max(r[1],r[0])):5d} +{int(r[3]-r[2]):5d} +{int(r[4]-r[3]):5d}")
    print(f"{s-2:3d}: " + " | ".join(row))
