"""Long run of the C2 engine: finite, decreasing loss over many steps; throughput.   python tools/soak.py [steps]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
B, D = 4096, 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 1000000)] + rng.normal(0, 1, (1000000, D))).astype(np.float32)
p = em.Parameters(n_steps=steps, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6))
eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
eng.set_data(data)
eng.run_steps(100)
torch.cuda.synchronize()
t0 = time.perf_counter()
done = 100
while done < steps:
    k = min(5000, steps - done)
    eng.run_steps(k)
    done += k
torch.cuda.synchronize()
dt = time.perf_counter() - t0
loss = eng.losses()
assert np.isfinite(loss).all(), "non-finite loss"
terms = eng.loss_terms()
print(f"{steps} steps: {dt / (steps - 100) * 1e6:.1f} us/step; loss {loss[0]:.3f} -> {loss[99]:.3f} -> {loss[len(loss) // 2]:.4f} -> {loss[-1]:.4f}; "
      f"min {loss.min():.4f}; last terms " + ", ".join(f"{k}={v[-1]:.4g}" for k, v in terms.items()))
z = eng.encode(data[:4096])
print("latent extent", np.abs(z).max(), "finite", np.isfinite(z).all())
