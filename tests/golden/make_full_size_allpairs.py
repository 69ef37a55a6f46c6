"""Golden numbers of the all-pairs (sketch-map) cost at the FULL batch sizes of BASELINE.json's configurations 4 and 5 -- run in the
build container only (minutes of float64 CPU work); the GPU tests (tests/test_headline_gpu.py) compare the kernels with the
committed numbers instead of with themselves.

    python tests/golden/make_full_size_allpairs.py        ->  tests/golden/full_size_allpairs.npz

For each configuration: the seeded batch X (``make_batch`` -- numpy only, re-generated bit for bit by the test) and a seeded latent
Z; then, in float64 and chunked over rows, the direct-difference statement of oracle/direct_ops.py (SURVEY.md appendix B;
encodermap/loss_functions/loss_functions.py:351-367, encodermap/misc/distances.py:144-255, distances.py:86):

    cost  = scale / B^2 * sum_ij (S_h(dh_ij) - S_l(dl_ij))^2
    dZ_k  = 2 sum_j w_kj (z_k - z_j),  w_kj = -2 scale (S_h - S_l) S_l'(dl_kj) / (B^2 dl_kj)   (0 where dl_kj = 0)

stored as cost, |dZ|_2, the column sums of dZ (zero up to rounding) and 384 rows of dZ.  The periodic configuration walks the wrapped
differences feature chunk by feature chunk; the Euclidean one forms d^2 from the float64 Gram matrix of the CENTRED rows (relative
error ~1e-13 at these magnitudes) and is checked here against direct differences on 48 rows before anything is written.
"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import direct_ops as D  # noqa: E402

CONFIGS = {
    # name: (B, D, L, periodicity, sig, scale, seed)
    "C4": (8192, 512, 2, 2 * math.pi, (30.0, 12, 6, 1, 2, 6), 500.0, 4),
    "C5": (16384, 256, 2, float("inf"), (22.6, 12, 6, 1, 2, 6), 500.0, 5),
}
N_ROWS_KEPT = 384


def make_batch(name):
    """the seeded batch and latent of a configuration (numpy only: the GPU test calls this too)"""
    B, Dd, L, per, sig, scale, seed = CONFIGS[name]
    rng = np.random.default_rng(seed)
    if per == float("inf"):      # 16 blob centres N(0, 4), points = centre + N(0, 1)   
        c = rng.normal(0, 4, (16, Dd))
        x = c[rng.integers(0, 16, B)] + rng.normal(0, 1, (B, Dd))
    else:                        # three basins per dihedral, width 0.35, wrapped to (-pi, pi]
        c = rng.uniform(-math.pi, math.pi, (3, Dd))
        k = rng.integers(0, 3, (B, Dd))
        x = np.take_along_axis(c, k, 0) + 0.35 * rng.normal(0, 1, (B, Dd))
        x = np.remainder(x + math.pi, 2 * math.pi) - math.pi
    z = rng.normal(0, 1.2, (B, L))
    z[7] = z[3]                  # a coincident latent pair: the reference's dl == 0 mask
    rows = np.sort(rng.choice(B, N_ROWS_KEPT, replace=False))
    return x.astype(np.float32), z.astype(np.float32), rows


def sig_t(r, sig, a, b):
    return 1.0 - (1.0 + (2.0 ** (a / b) - 1.0) * (r / sig) ** a) ** (-b / a)


def sig_prime_t(r, sig, a, b):
    c = 2.0 ** (a / b) - 1.0
    u = r / sig
    return b * c * u ** (a - 1.0) / sig * (1.0 + c * u ** a) ** (-b / a - 1.0)


def full_size(name, chunk=64):
    B, Dd, L, per, sig, scale, _ = CONFIGS[name]
    x32, z32, rows = make_batch(name)
    X = torch.from_numpy(x32.astype(np.float64))
    Z = torch.from_numpy(z32.astype(np.float64))
    cost = 0.0
    dZ = torch.zeros(B, L, dtype=torch.float64)
    if per == float("inf"):
        Xc = X - X.mean(0, keepdim=True)
        nrm = (Xc * Xc).sum(1)
    t0 = time.time()
    for r0 in range(0, B, chunk):
        r1 = min(r0 + chunk, B)
        if per == float("inf"):
            d2 = nrm[r0:r1, None] + nrm[None, :] - 2.0 * (Xc[r0:r1] @ Xc.T)
            d2.clamp_(min=0.0)
            d2[torch.arange(r1 - r0), torch.arange(r0, r1)] = 0.0
            dh = d2.sqrt()
        else:
            s = torch.zeros(r1 - r0, B, dtype=torch.float64)
            for k0 in range(0, Dd, 64):
                d = (X[None, :, k0:k0 + 64] - X[r0:r1, None, k0:k0 + 64]).abs_()
                v = torch.minimum(d, per - d)
                v += (v == 0.0) * 1e-12              # distances.py:164-175
                s += (v * v).sum(2)
            dh = s.sqrt() + 1e-12
        dz = Z[r0:r1, None, :] - Z[None, :, :]
        dl = (dz * dz).sum(2).sqrt()
        diff = sig_t(dh, *sig[:3]) - sig_t(dl, *sig[3:])
        cost += float((diff * diff).sum())
        w = torch.where(dl > 0, -2.0 * scale * diff * sig_prime_t(dl, *sig[3:]) / (B * B) / dl.clamp(min=1e-300), torch.zeros(()).double())
        dZ[r0:r1] = 2.0 * (w.sum(1)[:, None] * Z[r0:r1] - w @ Z)
        if r0 % (chunk * 32) == 0:
            print(f"  {name}: rows {r0} / {B}  ({time.time() - t0:.0f} s)", flush=True)
    cost = scale * cost / (B * B)
    dZ = dZ.numpy()
    # cross-check against oracle/direct_ops.py's own functions on a sub-batch (all rows of the sub-batch, direct differences)
    sub = np.concatenate([rows[:45], [3, 7, 0]])
    l_sub, g_sub = D.sketchmap_fwd_bwd(x32[sub], z32[sub], per, sig, scale)
    Xs, Zs = X[sub], Z[sub]
    if per == float("inf"):
        dh_s = torch.from_numpy(D.pairwise_dist_direct(x32[sub]))
    else:
        dh_s = torch.from_numpy(D.pairwise_dist_periodic_direct(x32[sub], per))
    dl_s = torch.from_numpy(D.pairwise_dist_direct(z32[sub]))
    diff_s = sig_t(dh_s, *sig[:3]) - sig_t(dl_s, *sig[3:])
    assert abs(scale * float((diff_s ** 2).sum()) / len(sub) ** 2 - l_sub) < 1e-12 * max(1.0, abs(l_sub))
    if per == float("inf"):      # the Gram form of the full pass against direct differences, on the kept rows
        d2g = nrm[sub][:, None] + nrm[sub][None, :] - 2.0 * (Xc[sub] @ Xc[sub].T)
        d2g.clamp_(min=0.0)
        d2g.fill_diagonal_(0.0)
        err = float((d2g.sqrt() - dh_s).abs().max())
        assert err < 1e-9, err
        print(f"  {name}: Gram vs direct distances on {len(sub)} rows: max |diff| = {err:.2e}")
    return {"cost": cost, "dz_norm": float(np.sqrt((dZ * dZ).sum())), "dz_colsum": dZ.sum(0), "rows": rows, "dz_rows": dZ[rows],
            "dz_absmax": float(np.abs(dZ).max())}


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count() or 8)
    out = {}
    for name in CONFIGS:
        r = full_size(name)
        print(name, {k: (v if np.ndim(v) == 0 else np.shape(v)) for k, v in r.items()}, flush=True)
        for k, v in r.items():
            out[f"{name}_{k}"] = np.asarray(v)
    np.savez(os.path.join(ROOT, "tests", "golden", "full_size_allpairs.npz"), **out)
