import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import encodermap_amd as em
from encodermap_amd.features import featurize_backbone
rng = np.random.default_rng(3)
N, A = 1500, 3 * 76
st = rng.normal(size=(N, A, 3)); st /= np.linalg.norm(st, axis=2, keepdims=True)
cv = featurize_backbone(np.cumsum(1.4 * st, axis=1).astype(np.float32))
def make(fuse):
    p = em.ADCParameters(n_steps=300, batch_size=256, seed=5, cartesian_pwd_start=1, cartesian_pwd_step=3, use_backbone_angles=True,
                         cartesian_dist_sig_parameters=(40.0, 10, 5, 1, 2, 5), cartesian_cost_scale_soft_start=(100, 200))
    eng = em.AngleDihedralCartesianEncoderMap(cv, p, read_only=True)._engine
    eng._prefetch = fuse == "1"
    return eng
e1, e0 = make("1"), make("0")
names = ["dih_pred", "ang_pred", "plane", "xyz", "inp_pair", "out_pair", "split_pair", "dpair", "dxyz", "dZ", "dY", "w"]
for nsteps in (1, 2, 3):
    e1.run_steps(nsteps); e0.run_steps(nsteps)
    torch.cuda.synchronize()
    bad = [n for n in names if not torch.equal(getattr(e1, n), getattr(e0, n))]
    print("after step", e1.current_step(), "differ:", bad, "losses", e1.losses()[-1], e0.losses()[-1])
    for n in bad[:4]:
        x, y = getattr(e1, n), getattr(e0, n)
        d = (x != y).nonzero()
        print("   ", n, x.shape, "n differing", len(d), "first", d[0].tolist())
