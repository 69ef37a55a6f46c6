import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine
INF = float("inf")
for B, D, per in ((256, 10, 2 * np.pi), (512, 128, INF), (1024, 128, INF), (256, 128, INF), (512, 32, 2 * np.pi)):
    rng = np.random.default_rng(1)
    data = (rng.uniform(-np.pi, np.pi, (50000, D)) if per < INF else rng.normal(0, 4, (16, D))[rng.integers(0, 16, 50000)] + rng.normal(0, 1, (50000, D))).astype(np.float32)
    res = []
    for join in (True, False, True, False):
        SequentialEngine._join_bwd = join
        p = em.Parameters(periodicity=per, batch_size=B, dist_sig_parameters=(float(np.sqrt(2 * D)), 12, 6, 1, 2, 6), n_steps=4000, seed=3)
        eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
        eng.set_data(data)
        eng.run_steps(40)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run_steps(400)
        torch.cuda.synchronize()
        res.append(((time.perf_counter() - t0) / 400 * 1e6, eng._bwd_ran, float(eng.losses()[-1])))
    print(f"B={B} D={D} periodic={per < INF}: join {res[0][0]:.2f} / {res[2][0]:.2f} us (ran {res[0][1]})  separate {res[1][0]:.2f} / {res[3][0]:.2f}  losses equal: {res[0][2] == res[1][2]}", flush=True)
