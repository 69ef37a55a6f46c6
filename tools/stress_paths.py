"""Random shapes through the step plans of round 3: (a) tiles + encoder backward aboard the fused launch against the separate launches --
bit for bit (the one-launch tail off on both sides; EMX_TCHAIN=0: a statement about the row-block chain kernel); (b) the one-launch
tail without slice partials against the two launches -- loss trajectories to 5e-4; (c) round 5: the default choice of the fused
launch's kernel (the transposed chain where a network qualifies) against the row-block kernel -- to 5e-4 as well.
python tools/stress_paths.py [n_shapes] [seed]"""
import os, sys
os.environ.setdefault("EMX_OVERLAP_MAX_B", "1536")
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import encodermap_amd as em
from encodermap_amd.engine import SequentialEngine

n_shapes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
INF = float("inf")
bad = transposed = 0
for it in range(n_shapes):
    periodic = rng.random() < 0.4
    B = int(rng.choice([5, 16, 17, 48, 100, 250, 256, 300, 512, 700, 1000, 1024, 1500, 2048, 3000]))
    D = int(rng.choice([3, 4, 7, 10, 16, 31, 32, 33, 64, 100, 128] if periodic else [4, 7, 12, 32, 33, 64, 100, 128, 256]))
    if periodic and B > 1024:
        B = 512
    depth = int(rng.integers(1, 4))
    neurons = [int(rng.choice([8, 20, 33, 64, 128, 200])) for _ in range(depth)] + [2]
    acts = [""] + ["tanh"] * depth + [""]
    steps = 6
    data = (rng.uniform(-np.pi, np.pi, (4000, D)) if periodic else rng.normal(0, 3, (8, D))[rng.integers(0, 8, 4000)] + rng.normal(0, 1, (4000, D))).astype(np.float32)
    idx = torch.from_numpy(rng.integers(0, 4000, (steps, B)).reshape(-1)).cuda()
    sig = (float(np.sqrt(2 * D)) if not periodic else 2.0, 12, 6, 1, 2, 6)

    def run(overlap, join, local):
        SequentialEngine._overlap_tiles, SequentialEngine._join_bwd, SequentialEngine._local_wgrad = overlap, join, local
        p = em.Parameters(periodicity=2 * np.pi if periodic else INF, batch_size=B, dist_sig_parameters=sig, n_steps=steps, n_neurons=neurons,
                          activation_functions=acts, seed=5)
        e = em.EncoderMap(p, data, read_only=True)
        eng = e._engine
        eng.run_steps(steps, idx)
        l = eng.losses()
        assert (eng.sk_sync.cpu().numpy() == 0).all() and (eng.wg_sync.cpu().numpy() == 0).all(), "sync words not clean"
        return l, eng.w.cpu().numpy(), (eng._tiles_ran, eng._bwd_ran, eng._tail_ran)

    try:
        os.environ["EMX_TCHAIN"] = "0"
        a, wa, fa = run(True, True, False)
        b, wb, fb = run(False, False, False)
        c, wc, fc = run(True, True, True)
        del os.environ["EMX_TCHAIN"]
        n0 = em._lib.load().emx_transposed_chain_count()
        d, wd, fd = run(False, False, False)
        transposed += em._lib.load().emx_transposed_chain_count() > n0
        ok1 = np.array_equal(a, b) and np.array_equal(wa, wb)
        ok2 = np.isfinite(c).all() and np.allclose(c, b, rtol=5e-4, atol=1e-6)
        ok3 = np.isfinite(d).all() and np.allclose(d, b, rtol=5e-4, atol=1e-6)
        tag = "ok" if ok1 and ok2 and ok3 else "MISMATCH"
        bad += tag != "ok"
    except Exception as exc:   # noqa: BLE001
        tag, fa, fc = f"ERROR {type(exc).__name__}: {exc}", None, None
        bad += 1
    print(f"{it:3d} B={B:5d} D={D:4d} periodic={periodic!s:5} n_neurons={neurons}: {tag}  aboard/join ran {fa}  tail ran {fc[2] if fc else None}", flush=True)
print("failures:", bad, " shapes whose default fused launch was the transposed chain:", transposed)
sys.exit(1 if bad else 0)
