"""Small-batch soak: the default EncoderMap configuration (C1) and the ADC model, a few thousand steps each, twice from the same
seed -- the two loss histories must be finite, decreasing and IDENTICAL bit for bit (every reduction is fixed-order).
    python tools/soak_small.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import encodermap_amd as em
from encodermap_amd.features import featurize_backbone

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000


def run_c1():
    x = np.random.default_rng(0).random((10000, 10), dtype=np.float32) * 2 * np.pi - np.pi
    e = em.EncoderMap(em.Parameters(n_steps=steps, seed=5), x, read_only=True)
    e._engine.run_steps(steps)
    return e._engine.losses()


def run_adc():
    rng = np.random.default_rng(3)
    N, A = 5000, 3 * 40
    st = rng.normal(size=(N, A, 3))
    st /= np.linalg.norm(st, axis=2, keepdims=True)
    cv = featurize_backbone(np.cumsum(1.4 * st, axis=1).astype(np.float32))
    p = em.ADCParameters(n_steps=steps, batch_size=256, seed=5, cartesian_pwd_start=1, cartesian_pwd_step=3, use_backbone_angles=True,
                         cartesian_dist_sig_parameters=(40.0, 10, 5, 1, 2, 5), dihedral_cost_reference=1.5, angle_cost_reference=0.3,
                         cartesian_cost_reference=8.0, cartesian_cost_scale_soft_start=(100, 400), track_clashes=True, track_RMSD=True)
    m = em.AngleDihedralCartesianEncoderMap(cv, p, read_only=True)
    m._engine.run_steps(steps)
    return m._engine.losses()


for name, fn in (("C1", run_c1), ("ADC", run_adc)):
    a, b = fn(), fn()
    torch.cuda.synchronize()
    ok = np.isfinite(a).all() and np.array_equal(a, b) and a[-50:].mean() < a[:50].mean()
    print(f"{name}: {steps} steps, loss {a[0]:.4f} -> {a[-1]:.4f}, finite {np.isfinite(a).all()}, identical reruns {np.array_equal(a, b)}", flush=True)
    if not ok:
        sys.exit(1)
print("ok")
