"""Step time of the Euclidean C2-shaped engine at small batches, for one choice of step plan per process (the switches are read at
import): default (tiles and the encoder's backward units aboard the fused launch up to EMX_OVERLAP_MAX_B rows) against
EMX_OVERLAP_TILES=0 (separate launches: the fused launch is then the transposed chain).   python tools/time_plans_small_b.py [B ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine
D = 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 200000)] + rng.normal(0, 1, (200000, D))).astype(np.float32)
for B in [int(v) for v in sys.argv[1:]] or [256, 512, 1024, 1536, 2048]:
    p = em.Parameters(n_steps=4000, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), seed=3)
    eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
    eng.set_data(data)
    eng.run_steps(200)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        eng.run_steps(1024)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 1024 * 1e6)
    print(f"EMX_OVERLAP_TILES={os.environ.get('EMX_OVERLAP_TILES', '')} EMX_TCHAIN={os.environ.get('EMX_TCHAIN', '')} B {B:5d}: {best:7.2f} us/step  tiles aboard {eng._tiles_ran} bwd aboard {eng._bwd_ran} "
          f"loss {eng.losses()[-1]:.4f}", flush=True)
