"""Fixed cost of one run_steps() call: wall time of run_steps(k) + synchronize, repeated, against k x the steady step time (C2 shape).
python tools/time_short_runs.py [k=20]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine

k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, D = 4096, 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 200000)] + rng.normal(0, 1, (200000, D))).astype(np.float32)
p = em.Parameters(n_steps=100000, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), seed=3)
eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
eng.set_data(data)
eng.run_steps(5)          # (the driver's shape: 5 warm-up steps, then the timed calls)
torch.cuda.synchronize()
steady = float(os.environ.get("STEADY_US", "73.4")) * 1e-6
ts = []
for _ in range(200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_steps(k)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0, t_enq))
print("first calls (us):", [round(t[0] * 1e6) for t in ts[:8]])
ts = np.array(ts[20:])
print(f"steady {steady * 1e6:.2f} us/step; run_steps({k}) + sync: median {np.median(ts[:, 0]) * 1e6:.1f} us = {np.median(ts[:, 0]) / k * 1e6:.2f} us/step "
      f"(fixed {np.median(ts[:, 0]) * 1e6 - k * steady * 1e6:.1f} us; host enqueue time {np.median(ts[:, 1]) * 1e6:.1f} us)")
