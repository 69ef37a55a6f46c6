"""Soak of the ADC step as round 3 leaves it (five launches: per-frame halves in one, split reduction in one, one-launch tail with the
next head): many steps twice from one seed -- finite, bit-identical, counters clean.   python tools/soak_adc.py [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
import bench   # (its synthetic backbone: a random-walk chain featurised by the product's own kernel)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
data = bench.make_backbone(dict(n_res=76), 4000)
runs = []
for rep in range(2):
    p = em.ADCParameters(n_steps=steps, batch_size=256, cartesian_pwd_start=1, cartesian_pwd_step=3, cartesian_dist_sig_parameters=(20.0, 10, 5, 1, 2, 5),
                         cartesian_cost_scale_soft_start=(50, 500), seed=3)
    e = em.AngleDihedralCartesianEncoderMap(data, p, read_only=True)
    eng = e._engine
    eng.run_steps(steps)
    l = eng.losses()
    assert eng._tail_ran and (eng.wg_sync.cpu().numpy() == 0).all() and int(eng.step) == steps
    runs.append(l)
assert np.isfinite(runs[0]).all() and np.array_equal(runs[0], runs[1])
print(f"ADC 228 atoms, B=256: {steps} steps x 2, loss {runs[0][0]:.3f} -> {runs[0][-1]:.3f}, identical, counters clean")
