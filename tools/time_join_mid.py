"""Batches between 1024 and 4096 rows: the fused launch with tiles and encoder backward aboard against the separate launches
(EMX_OVERLAP_MAX_B lifted).   python tools/time_join_mid.py"""
import os, sys, time
os.environ["EMX_OVERLAP_MAX_B"] = "4096"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine
INF = float("inf")
for B, D in ((1024, 256), (1280, 256), (1536, 256), (1536, 64), (1536, 32)) if os.environ.get("MID") == "2" else ((1536, 128), (1792, 128), (2048, 128), (2048, 64), (2048, 256), (2560, 128)) if os.environ.get("MID") else ((1024, 128), (1280, 128), (1536, 128), (2048, 128), (2048, 64), (3072, 128), (4096, 128)):
    rng = np.random.default_rng(1)
    data = (rng.normal(0, 4, (16, D))[rng.integers(0, 16, 100000)] + rng.normal(0, 1, (100000, D))).astype(np.float32)
    res = []
    for aboard in (True, False, True, False):
        SequentialEngine._overlap_tiles = aboard
        SequentialEngine._join_bwd = aboard
        p = em.Parameters(periodicity=INF, batch_size=B, dist_sig_parameters=(float(np.sqrt(2 * D)), 12, 6, 1, 2, 6), n_steps=4000, seed=3)
        eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
        eng.set_data(data)
        eng.run_steps(40)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_steps(400)
        torch.cuda.synchronize()
        res.append(((time.perf_counter() - t0) / 400 * 1e6, eng._bwd_ran))
    print(f"B={B} D={D}: aboard {res[0][0]:.2f} / {res[2][0]:.2f} us (ran {res[0][1]})   separate {res[1][0]:.2f} / {res[3][0]:.2f}", flush=True)
