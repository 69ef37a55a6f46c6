"""The all-pairs tile pass alone, timed inside a hipGraph of many launches (the eager loop of time_sk.py is host-bound near 15 us per
call), for a list of batch sizes; cost and latent gradient checked against a float64 direct computation on the device.
Environment switches of the library (EMX_SK_W16_MIN_B, EMX_SK_BIG_MIN_B, ...) are read once per process: one process per variant.
usage: python tools/time_sk_graph.py [D] [B ...]      run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from encodermap_amd import _lib

_lib.load()
D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
Bs = [int(v) for v in sys.argv[2:]] or [4096]
sig = (16.0 * (D / 128) ** 0.5, 12, 6, 1, 2, 6)
for B in Bs:
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    cen = torch.randn(16, D, generator=g, device="cuda") * 4
    X = cen[torch.randint(0, 16, (B,), generator=g, device="cuda")] + torch.randn(B, D, generator=g, device="cuda")
    Z = torch.randn(B, 2, generator=g, device="cuda")
    ws = torch.zeros(_lib.sketchmap_ws_bytes(B, 2) // 4, device="cuda")
    split = torch.zeros(_lib.sketchmap_split_bytes(B, D), dtype=torch.uint8, device="cuda")
    _lib.sketchmap_split(X, split)
    dist = torch.zeros(1, device="cuda")
    dZ = torch.zeros(B, 2, device="cuda")
    for _ in range(3):
        _lib.sketchmap_tiles(X, Z, float("inf"), sig, 0, B, ws, split)
    _lib.sketchmap_finish(B, D, Z, sig, 1.0, 0.0, 0, B, dist, None, dZ, ws)
    torch.cuda.synchronize()
    # float64 direct form, row-chunked
    Xd, Zd = X.double(), Z.double()
    cost = torch.zeros((), dtype=torch.float64, device="cuda")
    gz = torch.zeros(B, 2, dtype=torch.float64, device="cuda")

    def sg(r2, s, a, b):      # sigmoid of the distance as a function of r^2, and its derivative with respect to r^2
        c = 2.0 ** (a / b) - 1.0
        v = r2 / (s * s)
        t = v ** (a / 2)
        q = 1 + c * t
        return 1 - q ** (-b / a), (b / a) * q ** (-b / a - 1) * c * (a / 2) * v ** (a / 2 - 1) / (s * s)

    for s0 in range(0, B, 1024):
        xs, zs = Xd[s0:s0 + 1024], Zd[s0:s0 + 1024]
        r2h = (xs * xs).sum(1)[:, None] + (Xd * Xd).sum(1)[None, :] - 2 * xs @ Xd.T
        r2h.clamp_(min=0)
        dz = zs[:, None, :] - Zd[None, :, :]
        r2l = (dz * dz).sum(2)
        sh, _ = sg(r2h, sig[0], sig[1], sig[2])
        sl, dsl = sg(r2l, sig[3], sig[4], sig[5])
        f = sh - sl
        cost += (f * f).sum()
        gz[s0:s0 + 1024] += (2 * f * (-dsl))[:, :, None].mul(2 * dz).sum(1) * 2     # (each pair enters as (i, j) and (j, i))
    cost /= B * B
    gz /= B * B
    err_c = abs(float(dist) - float(cost)) / float(cost)
    err_g = float((dZ.double() - gz).abs().max() / gz.abs().max())
    n = 50
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            _lib.sketchmap_tiles(X, Z, float("inf"), sig, 0, B, ws, split)
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    print(f"B {B} D {D}: tile pass {min(ts):.2f} us (median {sorted(ts)[2]:.2f})   cost rel err {err_c:.2e}   dZ err / max {err_g:.2e}", flush=True)
