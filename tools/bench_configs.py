"""Step time of the other BASELINE.json configurations on ONE GPU (they are parity-test cases, not bench lines):
C1 (defaults: 10000x10, B=256, periodic), C4 shape (D=512 periodic, B=8192; 2e5 rows resident instead of 2e6), C5 shape
(D=256 Euclidean, B=16384; 1e5 rows).   python tools/bench_configs.py [C1 C4 C5 ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import encodermap_amd as em  # noqa: E402
from encodermap_amd.engine import SequentialEngine  # noqa: E402

CFG = {
    "C1": dict(N=10000, D=10, kw=dict()),
    "C2": dict(N=200000, D=128, kw=dict(periodicity=float("inf"), batch_size=4096, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6))),
    "C4": dict(N=200000, D=512, kw=dict(periodicity=2 * np.pi, batch_size=8192)),
    "C5": dict(N=100000, D=256, kw=dict(periodicity=float("inf"), batch_size=16384, dist_sig_parameters=(22.6, 12, 6, 1, 2, 6))),
}
which = [a for a in sys.argv[1:] if a in CFG] or ["C1", "C4", "C5"]
for name in which:
    c = CFG[name]
    rng = np.random.default_rng(1)
    if np.isfinite(c["kw"].get("periodicity", 2 * np.pi)):
        data = rng.uniform(-np.pi, np.pi, (c["N"], c["D"])).astype(np.float32)
    else:
        cen = rng.normal(0, 4, (16, c["D"]))
        data = (cen[rng.integers(0, 16, c["N"])] + rng.normal(0, 1, (c["N"], c["D"]))).astype(np.float32)
    p = em.Parameters(n_steps=4000, **c["kw"])
    eng = SequentialEngine(c["D"], p, with_distance=True, device="cuda:0")
    eng.set_data(data)
    eng.run_steps(20)
    torch.cuda.synchronize()
    steps = 200
    t0 = time.perf_counter()
    eng.run_steps(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    B = p.batch_size
    W = c["D"] * B * (B - 1) / 2
    f = 3 if not np.isfinite(p.periodicity) else 5
    print(f"{name}: D={c['D']} B={B} periodic={np.isfinite(p.periodicity)}  {dt * 1e6:.1f} us/step  {B / dt:.3e} samples/s  "
          f"pair work {f * W / 1e9:.1f} GFLOP -> whole step at {f * W / dt / 1e12:.1f} TFLOP/s of pair work", flush=True)
    del eng
