"""Is the slower start of a run the DEVICE (first milliseconds of load) or the TRAINING STATE (early weights)?  The same 25 steps
(5 + 20 timed) three times from the SAME initial state (weights, moments, counters and shuffle stream restored in between), then 20
steps from a state 2000 steps into training.   python tools/time_early_steps.py"""
import os, sys, time, copy
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine

B, D = 4096, 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 200000)] + rng.normal(0, 1, (200000, D))).astype(np.float32)
p = em.Parameters(n_steps=100000, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), seed=3)
eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
eng.set_data(data)
idx = torch.from_numpy(rng.integers(0, 200000, (25, B)).reshape(-1)).cuda()
sd0 = {k: (v.clone() if torch.is_tensor(v) else copy.deepcopy(v)) for k, v in eng.state_dict().items()}


def timed(n_warm, n, block):
    eng.run_steps(n_warm, block[: n_warm * B])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_steps(n, block[n_warm * B: (n_warm + n) * B])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for rep in range(4):
    eng.load_state_dict(sd0)
    print(f"from the initial state, repetition {rep}: {timed(5, 20, idx):.2f} us/step", flush=True)
eng.load_state_dict(sd0)
eng.run_steps(2000)
torch.cuda.synchronize()
for rep in range(3):
    print(f"2000+ steps into training: {timed(5, 20, idx):.2f} us/step", flush=True)
eng.load_state_dict(sd0)
print(f"back at the initial state: {timed(5, 20, idx):.2f} us/step")
