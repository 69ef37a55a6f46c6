"""The ADC step with the fused frame pass against the single stages: same seed, loss histories compared bit for bit."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    import encodermap_amd as em
    from encodermap_amd.features import featurize_backbone
    rng = np.random.default_rng(3)
    N, A = 1500, 3 * 76
    st = rng.normal(size=(N, A, 3)); st /= np.linalg.norm(st, axis=2, keepdims=True)
    cv = featurize_backbone(np.cumsum(1.4 * st, axis=1).astype(np.float32))
    p = em.ADCParameters(n_steps=300, batch_size=256, seed=5, cartesian_pwd_start=1, cartesian_pwd_step=3, use_backbone_angles=sys.argv[2] == "1",
                         cartesian_dist_sig_parameters=(40.0, 10, 5, 1, 2, 5), cartesian_cost_scale_soft_start=(100, 200))
    m = em.AngleDihedralCartesianEncoderMap(cv, p, read_only=True)
    m._engine.run_steps(300)
    np.save(sys.argv[3], m._engine.losses())
    sys.exit(0)
import numpy as np
for ang in ("1", "0"):
    out = []
    for fuse in ("1", "0", "1"):
        f = f"/tmp/adc_fuse_{ang}_{len(out)}.npy"
        subprocess.run([sys.executable, __file__, "child", ang, f], check=True, env={**os.environ, "EMX_ADC_FUSE_FRAME": fuse})
        out.append(np.load(f))
    print(f"use_backbone_angles={ang}: fused rerun identical {np.array_equal(out[0], out[2])}, fused == stages {np.array_equal(out[0], out[1])}, "
          f"first difference at step {int(np.argmax(out[0] != out[1])) if not np.array_equal(out[0], out[1]) else -1}, last losses {out[0][-1]:.6f} {out[1][-1]:.6f}")
