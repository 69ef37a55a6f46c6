"""encode() of a 1e6 x 128 training set: from the HBM-resident copy vs from the host array; run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import encodermap_amd as em

N, D = 1_000_000, 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D)).astype(np.float32)
x = cen[rng.integers(0, 16, N)] + rng.standard_normal((N, D), dtype=np.float32)
e = em.EncoderMap(em.Parameters(periodicity=float("inf"), batch_size=4096, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6)), x, read_only=True)
for label, f in (("resident", lambda: e.encode()), ("host array", lambda: e.encode(x))):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        z = f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"encode {label}: {dt * 1e3:.1f} ms for {N} rows = {N / dt / 1e6:.1f} M rows/s", flush=True)
