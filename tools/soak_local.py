"""Soak of the one-launch step tail without slice partials (emx_mlp_wgrad_adam_local): many steps at several batch sizes and widths,
twice from one seed -- finite, decreasing, bit-identical, the launch's counters back at zero, the epilogue never gave up waiting.
python tools/soak_local.py [steps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(1)
for B, D, per in ((256, 10, 2 * np.pi), (256, 32, float("inf")), (1000, 100, float("inf")), (2048, 128, float("inf")), (4096, 128, float("inf"))):
    cen = rng.normal(0, 4, (16, D))
    data = (cen[rng.integers(0, 16, 100000)] + rng.normal(0, 1, (100000, D))).astype(np.float32)
    if per < float("inf"):
        data = rng.uniform(-np.pi, np.pi, (100000, D)).astype(np.float32)
    runs = []
    for rep in range(2):
        p = em.Parameters(periodicity=per, batch_size=B, dist_sig_parameters=(float(np.sqrt(2 * D)), 12, 6, 1, 2, 6), n_steps=steps, seed=11)
        e = em.EncoderMap(p, data, read_only=True)
        e._engine.run_steps(steps)
        l = e._engine.losses()
        assert e._engine._tail_ran and e._engine._local_ok and (e._engine.wg_sync.cpu().numpy() == 0).all()
        assert int(e._engine.step) == steps and int(e._engine.hstep) == steps
        runs.append(l)
    assert np.isfinite(runs[0]).all() and np.array_equal(runs[0], runs[1]) and runs[0][-100:].mean() < runs[0][:100].mean()
    print(f"B={B} D={D}: {steps} steps x 2, loss {runs[0][0]:.3f} -> {runs[0][-1]:.3f}, identical, counters clean", flush=True)
