"""Runs the sketch-map tile kernel alone (C2 shape by default) for profiling; run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from encodermap_amd import _lib
if os.environ.get("EMX_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["EMX_LIB"])
_lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
per = float(sys.argv[3]) if len(sys.argv) > 3 else float("inf")
g = torch.Generator(device="cuda"); g.manual_seed(0)
if per == float("inf"):
    cen = torch.randn(16, D, generator=g, device="cuda") * 4
    X = cen[torch.randint(0, 16, (B,), generator=g, device="cuda")] + torch.randn(B, D, generator=g, device="cuda")
    sig = (16.0, 12, 6, 1, 2, 6)
else:
    X = (torch.rand(B, D, generator=g, device="cuda") - 0.5) * per
    sig = (per * (D ** 0.5) * 0.2, 12, 6, 1, 2, 6)
Z = torch.randn(B, 2, generator=g, device="cuda")
ws = torch.zeros(_lib.sketchmap_ws_bytes(B, 2) // 4, device="cuda")
split = None
if per == float("inf") and os.environ.get("EMX_TIME_SK_F32", "0") != "1":   # the engine's path: Gram products on the f16 cores
    split = torch.zeros(_lib.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
    _lib.sketchmap_split(X, split)
reps = 30
span = float(os.environ["EMX_TIME_SK_SPAN"]) if os.environ.get("EMX_TIME_SK_SPAN") else None   # data_span: selects the periodic tile form
_tiles = _lib.sketchmap_tiles
_lib.sketchmap_tiles = lambda *a: _tiles(*a, data_span=span)
for _ in range(3): _lib.sketchmap_tiles(X, Z, per, sig, 0, B, ws, split)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(reps): _lib.sketchmap_tiles(X, Z, per, sig, 0, B, ws, split)
e1.record(); torch.cuda.synchronize()
print("B", B, "D", D, "per", per, "tile kernel us", e0.elapsed_time(e1) / reps * 1e3)
