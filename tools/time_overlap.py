"""A/B of the fused dense-layer launch with and without the all-pairs tiles aboard (emx_mlp_forward_backward_tiles), C2 shape by
default:  python tools/time_overlap.py [B] [D]
Prints the launch times (HIP events around back-to-back launches on a live batch) and the step times of both plans."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import encodermap_amd as em  # noqa: E402
from encodermap_amd import _lib  # noqa: E402
from encodermap_amd.engine import SequentialEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 128
INF = float("inf")
os.environ.setdefault("EMX_OVERLAP_MAX_B", "1000000")   # (the engine limits the tiles-aboard plan to small batches: lifted for the A/B)
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 200000)] + rng.normal(0, 1, (200000, D))).astype(np.float32)
sig = (float(np.sqrt(2 * D)), 12, 6, 1, 2, 6)


def timed(fn, n=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def step_time(overlap, steps=400):
    SequentialEngine._overlap_tiles = overlap
    p = em.Parameters(periodicity=INF, batch_size=B, dist_sig_parameters=sig, n_steps=4000, seed=3)
    eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
    eng.set_data(data)
    eng.run_steps(40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_steps(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e6
    assert eng._tiles_ran == overlap, (eng._tiles_ran, overlap)
    eng.check_sync()
    return dt, eng


for rep in range(2):
    t_sep, eng = step_time(False)
    Z = eng.h[eng.layout.n_enc - 1]
    eng._overlap_tiles = False
    t_fused = timed(eng._forward)
    t_tiles = timed(lambda: _lib.sketchmap_tiles(eng.xb, Z, INF, sig, 0, B, eng.sk_ws, eng.split_ws))
    eng._overlap_tiles = True
    t_both = timed(eng._forward)
    eng.check_sync()
    del eng
    t_ovl, eng2 = step_time(True)
    del eng2
    print(f"B={B} D={D}: fused launch {t_fused:.1f} us, tile launch {t_tiles:.1f} us (sum {t_fused + t_tiles:.1f}), fused launch with the tiles aboard "
          f"{t_both:.1f} us | step: separate launches {t_sep:.1f} us, tiles aboard {t_ovl:.1f} us", flush=True)
