"""tests/test_adc_gpu.py::test_adc_free_running_drift_envelope over several seeds: the spread of the drift ratios (the trajectory
is chaotic: 2-3 % deviation of the float32 oracle from the float64 one within 10 steps)."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import encodermap_amd as em
from oracle import reference_ops as R
from tests.test_adc_gpu import CASES, weights_for
from tests.test_training_gpu import drift_ratio
name, kw, n_res, n_side, B = CASES[3]
for seed in [n_res, n_res + 1, n_res + 2, n_res + 3]:
    rng = np.random.default_rng(seed)
    data = R.synthetic_backbone(rng, 700, n_res, n_side)
    n_steps = 10
    p = em.ADCParameters(n_steps=n_steps, batch_size=B, **kw)
    A = 3 * n_res
    w0 = weights_for(em, p, 2 * (A - 3))
    idx = rng.integers(0, 700, (n_steps, B))
    e = em.AngleDihedralCartesianEncoderMap(data, p, model=[w.copy() for w in w0], read_only=True)
    e._engine.run_steps(n_steps, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    got, got_terms = e._engine.losses(), e._engine.loss_terms()
    runs = {dt: R.adc_train_steps(R.ADCOracle(A, n_side, p.to_dict(), w0, dtype=dt), data, idx) for dt in (torch.float64, torch.float32)}
    (f64, t64), (f32, t32) = runs[torch.float64], runs[torch.float32]
    ratios = {"loss": drift_ratio(got, f64, f32, rel=2e-4)}
    for k in t64[0]:
        ratios[k] = drift_ratio(got_terms[k], [t[k] for t in t64], [t[k] for t in t32], rel=2e-4)
    print("seed", seed, {k: round(v, 2) for k, v in ratios.items() if v != 0}, "final rel. deviation engine %.2e oracle-f32 %.2e" % (abs(got[-1] - f64[-1]) / abs(f64[-1]), abs(f32[-1] - f64[-1]) / abs(f64[-1])), flush=True)
