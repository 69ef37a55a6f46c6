import os, sys, time
os.environ["EMX_OVERLAP_MAX_B"] = "4096"
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine
INF = float("inf")
for B, D in ((512, 128), (1024, 128), (1536, 128), (2048, 128)):
    rng = np.random.default_rng(1)
    data = (rng.normal(0, 4, (16, D))[rng.integers(0, 16, 100000)] + rng.normal(0, 1, (100000, D))).astype(np.float32)
    p = em.Parameters(periodicity=INF, batch_size=B, dist_sig_parameters=(float(np.sqrt(2 * D)), 12, 6, 1, 2, 6), n_steps=4000, seed=3)
    eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
    eng.set_data(data)
    eng.run_steps(64)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        eng.run_steps(320)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 320 * 1e6)
    print(f"tile wgs per xcd {os.environ.get('EMX_TILE_WGS_PER_XCD', '32')}: B={B}: {best:.2f} us/step (aboard {eng._tiles_ran})", flush=True)
