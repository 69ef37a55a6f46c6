"""The persistent phase-alternating all-pairs kernel (sk_tile_f16_pp_kernel) against the one-tile-per-workgroup kernel it replaces
(sk_tile_f16_big_kernel): bit-identical cost and latent gradient, and the time of the tile launch alone.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from encodermap_amd import _lib as L

L.load()
INF = float("inf")
for B, D, sig0 in ((16384, 256, 22.6), (8192, 256, 22.6), (8300, 250, 22.0), (12290, 230, 21.0)):
    g = torch.Generator(device="cuda"); g.manual_seed(B)
    cen = torch.randn(16, D, generator=g, device="cuda") * 4
    x = cen[torch.randint(0, 16, (B,), generator=g, device="cuda")] + torch.randn(B, D, generator=g, device="cuda")
    z = torch.randn(B, 2, generator=g, device="cuda")
    sig = (sig0, 12, 6, 1, 2, 6)
    split = torch.zeros(L.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
    L.sketchmap_split(x, split)
    res = {}
    for pp in ("0", "1"):
        os.environ["EMX_SK_PP"] = pp
        ws = torch.zeros(L.sketchmap_ws_bytes(B, 2) // 4, dtype=torch.float32, device="cuda")
        dist, cen_l = torch.zeros(1, device="cuda"), torch.zeros(1, device="cuda")
        dZ = torch.full((B, 2), float("nan"), device="cuda")
        L.sketchmap_tiles(x, z, INF, sig, 0, B, ws, split)
        L.sketchmap_finish(B, D, z, sig, 500.0, 1e-4, 0, B, dist, cen_l, dZ, ws)
        torch.cuda.synchronize()
        for _ in range(3):
            L.sketchmap_tiles(x, z, INF, sig, 0, B, ws, split)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            L.sketchmap_tiles(x, z, INF, sig, 0, B, ws, split)
        e1.record()
        torch.cuda.synchronize()
        res[pp] = (float(dist), dZ.cpu().numpy(), e0.elapsed_time(e1) / 20 * 1e3)
    same = res["0"][0] == res["1"][0] and np.array_equal(res["0"][1], res["1"][1])
    rel = abs(res["0"][0] - res["1"][0]) / abs(res["0"][0])
    gerr = np.abs(res["0"][1] - res["1"][1]).max() / np.abs(res["0"][1]).max()
    print(f"B={B} D={D}: one-tile kernel {res['0'][2]:.1f} us, phase-alternating {res['1'][2]:.1f} us; cost {res['0'][0]:.9g} vs {res['1'][0]:.9g} "
          f"(rel {rel:.2e}), gradient max rel diff {gerr:.2e}, bit-identical: {same}", flush=True)
