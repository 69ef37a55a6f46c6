"""Times the optimizer launch of the C2 engine: with / without the next-batch gather and the operand-store update.
python tools/time_adam.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd import _lib
from encodermap_amd.engine import SequentialEngine, INF

B, D = 4096, 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 100000)] + rng.normal(0, 1, (100000, D))).astype(np.float32)
p = em.Parameters(n_steps=100, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6))
eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
eng.set_data(data)
eng.run_steps(10)
torch.cuda.synchronize()

def timeit(fn, reps=300):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

S = _lib.mlp_wgrad_slices(B)
nxt = _lib.make_next_batch(eng.data, eng.idx_block, eng.cursor, eng.B, INF, eng.xb, None, eng.split_ws)
def run(plan, nb):
    eng.cursor.zero_()
    _lib.adam_step_slices(eng.w, eng.wg_ws, S, eng.m, eng.v, eng.l2coef, 1e-3, 0.9, 0.999, 1e-7, 1.0, eng.step, eng.red_ws2, plan=plan, next_batch=nb,
                          bookkeeping_done=True)
for name, plan, nb in (("adam only", None, None), ("+ operand store", eng.plan, None), ("+ gather", None, nxt), ("all", eng.plan, nxt)):
    print(f"{name:18s} {timeit(lambda: run(plan, nb)):.2f} us (eager, includes a 4-byte memset launch)")
print(f"memset alone       {timeit(lambda: eng.cursor.zero_()):.2f} us")
