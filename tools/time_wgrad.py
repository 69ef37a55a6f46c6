"""Times the weight-gradient launch of the C2 engine with and without the scalar epilogue riding along, and the
stand-alone epilogue launch.   python tools/time_wgrad.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd import _lib
from encodermap_amd.engine import SequentialEngine, N_TERMS

B, D = 4096, 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 100000)] + rng.normal(0, 1, (100000, D))).astype(np.float32)
p = em.Parameters(n_steps=100, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6))
eng = SequentialEngine(D, p, with_distance=True, device="cuda:0")
eng.set_data(data)
eng.run_steps(10)
torch.cuda.synchronize()

def timeit(fn, reps=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

saved = [t.clone() for t in eng._state_tensors()]
eng._fold_finish = True
t_fold = timeit(eng._wgrad)
eng._fold_finish = False
t_plain = timeit(eng._wgrad)
jobs = eng._jobs_pre() + [(eng.red_ws2, _lib.EMX_REDUCE_WS, 1.0, None, eng._reg_slot, False)]
t_fin = timeit(lambda: _lib.finish_step(jobs, eng.terms, N_TERMS, eng.history, eng.term_history, eng.hstep, eng.cursor, eng.step))
print(f"wgrad + epilogue {t_fold:.2f} us   wgrad alone {t_plain:.2f} us   epilogue alone {t_fin:.2f} us  (back-to-back launches, eager)")
