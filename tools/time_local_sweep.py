"""Step time with the one-launch tail without slice partials (emx_mlp_wgrad_adam_local) against the two launches, over batch sizes
and widths:   python tools/time_local_sweep.py            (EMX_WGRAD_LOCAL_MAX_ROWS is lifted for the sweep)"""
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("EMX_WGRAD_LOCAL_MAX_ROWS", "1000000")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import encodermap_amd as em  # noqa: E402
from encodermap_amd.engine import SequentialEngine  # noqa: E402

INF = float("inf")


def step_time(B, D, neurons, local, steps=300):
    SequentialEngine._local_wgrad = local
    rng = np.random.default_rng(1)
    cen = rng.normal(0, 4, (16, D))
    data = (cen[rng.integers(0, 16, 100000)] + rng.normal(0, 1, (100000, D))).astype(np.float32)
    kw = dict(n_neurons=neurons, activation_functions=[""] + ["tanh"] * (len(neurons) - 1) + [""]) if neurons else {}
    p = em.Parameters(periodicity=INF, batch_size=B, dist_sig_parameters=(float(np.sqrt(2 * D)), 12, 6, 1, 2, 6), n_steps=4000, seed=3, **kw)
    eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
    eng.set_data(data)
    eng.run_steps(40)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter()
        eng.run_steps(steps)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    return best, eng._tail_ran


SHAPES = [(1024, 256, [256, 256, 2]), (4096, 256, [256, 256, 2]), (2048, 512, [512, 128, 2]), (512, 512, [512, 512, 2]), (4096, 512, [256, 128, 2])] if os.environ.get("SWEEP") == "wide" else [(8192, 256, None), (16384, 256, None), (8192, 512, [256, 128, 2]), (6144, 256, [256, 256, 2])] if os.environ.get("SWEEP") == "big" else None
for B, D, neurons in SHAPES or [(256, 128, None), (1024, 128, None), (2048, 128, None), (4096, 128, None), (6144, 128, None), (8192, 128, None),
                      (4096, 256, None), (4096, 128, [256, 256, 2]), (4096, 64, [64, 64, 2]), (1024, 256, [256, 256, 2])]:
    a, ran = step_time(B, D, neurons, True)
    b, _ = step_time(B, D, neurons, False)
    print(f"B={B:5d} D={D:4d} n_neurons={neurons or 'default'}: one launch {a:7.2f} us/step (ran: {ran})   two launches {b:7.2f}   difference {a - b:+6.2f}", flush=True)
