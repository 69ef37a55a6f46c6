"""The all-pairs tile pass at the C2 shape for latent widths 1 .. 24 (which tile kernel runs and how long):  python tools/time_sk_latent.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from encodermap_amd import _lib
_lib.load()
B, D = 4096, 128
g = torch.Generator(device="cuda"); g.manual_seed(0)
cen = torch.randn(16, D, generator=g, device="cuda") * 4
X = cen[torch.randint(0, 16, (B,), generator=g, device="cuda")] + torch.randn(B, D, generator=g, device="cuda")
split = torch.zeros(_lib.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
_lib.sketchmap_split(X, split)
for L in (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 24):
    Z = torch.randn(B, L, generator=g, device="cuda")
    ws = torch.zeros(_lib.sketchmap_ws_bytes(B, L) // 4, device="cuda")
    f = lambda: _lib.sketchmap_tiles(X, Z, float("inf"), (16.0, 12, 6, 1, 2, 6), 0, B, ws, split)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"latent width {L:2d}: {e0.elapsed_time(e1) / 50 * 1e3:8.1f} us")
