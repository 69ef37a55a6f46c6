"""How long the fused dense-layer launch takes INSIDE the step (eager launches, HIP events around it) against back-to-back
launches of it alone; under `rocprofv3 --kernel-trace --stats` the profiler's average of the same launches stands beside it:
    python tools/time_in_step.py [--alone 0|1] [--n 300]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, encodermap_amd as em

ap = argparse.ArgumentParser(); ap.add_argument("--alone", type=int, default=1); ap.add_argument("--n", type=int, default=300)
a = ap.parse_args()
cfg = bench.CONFIGS["C2"]
dev = "cuda:0"
p = em.Parameters(n_steps=4 * a.n, seed=7, **cfg["kw"])
B = cfg["kw"]["batch_size"]
data = bench.make_data(cfg, dev, 200000)
e = em.EncoderMap(p, np.zeros((B, cfg["D"]), "float32"), read_only=True, device=dev, use_graph=False)
eng = e._engine
eng.set_data(data)
eng.run_steps(20)
torch.cuda.synchronize()
pairs = []
inner = eng._forward
def timed_forward():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = inner(); e1.record()
    pairs.append((e0, e1))
    return r
eng._forward = timed_forward
eng.run_steps(a.n)
torch.cuda.synchronize()
eng._forward = inner
t = np.array([x.elapsed_time(y) * 1e3 for x, y in pairs])
print(f"fused launch inside eager steps, events around it: mean {t.mean():.2f} us, median {np.median(t):.2f} us over {len(t)} steps")
# an empty pair of events: what the events themselves cost
emp = []
for _ in range(200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record(); emp.append((e0, e1))
torch.cuda.synchronize()
te = np.array([x.elapsed_time(y) * 1e3 for x, y in emp])
print(f"empty event pair: mean {te.mean():.2f} us, median {np.median(te):.2f} us")
if a.alone:
    print(f"the same launch alone, back to back: {bench._time_launches(inner, a.n) * 1e6:.2f} us")
