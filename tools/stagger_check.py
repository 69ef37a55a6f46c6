"""EMX_SK_STAGGER sweep on the one-tile-per-workgroup macro-tile kernel at the C5 shape (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from encodermap_amd import _lib as L
L.load()
os.environ["EMX_SK_PP"] = "0"
B, D = 16384, 256
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.randn(B, D, generator=g, device="cuda") * 2
z = torch.randn(B, 2, generator=g, device="cuda")
sig = (22.6, 12, 6, 1, 2, 6)
split = torch.zeros(L.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
L.sketchmap_split(x, split)
ws = torch.zeros(L.sketchmap_ws_bytes(B, 2) // 4, dtype=torch.float32, device="cuda")
for rep in range(2):
    for stg in (0, 1, 2, 3, 4, 6, 8, 0):
        os.environ["EMX_SK_STAGGER"] = str(stg)
        for _ in range(3):
            L.sketchmap_tiles(x, z, float("inf"), sig, 0, B, ws, split)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            L.sketchmap_tiles(x, z, float("inf"), sig, 0, B, ws, split)
        e1.record(); torch.cuda.synchronize()
        print(f"stagger {stg} x s_sleep(127) per class: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
