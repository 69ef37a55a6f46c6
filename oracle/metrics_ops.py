"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's ADC training metrics.

Follows encodermap/callbacks/metrics.py of the reference:
  * kabsch_weighted_rmsd  -- ``kabsch_weighted`` (:71-145): weighted covariance of the two point sets, weighted centres
    removed, singular values of the 3x3 covariance, the smallest one negated when det(V) det(W) < 0,
    msd = (PSQ + QSQ) iw - 2 sum(S), clamped at 0.
  * rmsd_numpy            -- ``rmsd`` / ``rmsd_numpy`` (:147-172): weights = tile(WEIGHTS, n_atoms // 3) with
    WEIGHTS = [14.0067, 24.305, 24.305] (:63); with ``translate`` every atom is shifted by the mean of its own three
    coordinates (``a -= tile(mean(a, axis=-1))``, :149-151) -- restated as written, not as a centroid translation.
  * clash_counts          -- ``ADCClashMetric.update_state`` (:505-522): per frame the number of entries of the flat
    pairwise-distance vector below the clash distance (0.1 nm / 1 Angstrom).

Pinned on the known answers of the reference's tests/test_autoencoder.py:412-480 (tests/test_oracle_kat.py):
rmsd(c, c) = 0 within 1e-3, rmsd(c, c2) != 0, mean clashes 1.0 and 2.0 on the two literal coordinate sets.
Everything is float64 here; the reference computes in float32.
"""
import numpy as np

WEIGHTS = (14.0067, 24.305, 24.305)   # metrics.py:63


def kabsch_weighted_rmsd(P: np.ndarray, Q: np.ndarray, w: np.ndarray) -> float:
    """P, Q [n, 3]; w [n] per-atom weights.  metrics.py:71-145."""
    P = np.asarray(P, np.float64)
    Q = np.asarray(Q, np.float64)
    w = np.asarray(w, np.float64)
    iw = 3.0 / (3.0 * w.sum())                       # iw = 3 / sum of the [n, 3] weight array (:105-107)
    C = (P * w[:, None]).T @ Q                      # C[i, k] = sum_j P[j, i] Q[j, k] w[j]   (:109-112)
    cmp_, cmq = (P * w[:, None]).sum(0), (Q * w[:, None]).sum(0)
    psq = (P * P * w[:, None]).sum() - (cmp_ * cmp_).sum() * iw
    qsq = (Q * Q * w[:, None]).sum() - (cmq * cmq).sum() * iw
    C = (C - np.outer(cmp_, cmq) * iw) * iw
    V, S, Wt = np.linalg.svd(C)
    if np.linalg.det(V) * np.linalg.det(Wt) < 0.0:  # (:127-131)
        S[-1] = -S[-1]
    msd = (psq + qsq) * iw - 2.0 * S.sum()
    return float(np.sqrt(max(msd, 0.0)))


def rmsd_numpy(a: np.ndarray, b: np.ndarray, translate: bool = True, weights3=WEIGHTS) -> np.ndarray:
    """a, b [B, n, 3] -> [B].  metrics.py:147-172."""
    a = np.array(a, np.float64)
    b = np.array(b, np.float64)
    out = []
    for fa, fb in zip(a, b):
        w = np.tile(np.asarray(weights3, np.float64), fa.shape[0] // 3)
        if translate:
            fa = fa - fa.mean(axis=-1, keepdims=True)
            fb = fb - fb.mean(axis=-1, keepdims=True)
        out.append(kabsch_weighted_rmsd(fa, fb, w))
    return np.asarray(out)


def clash_counts(xyz: np.ndarray, clash_distance: float) -> np.ndarray:
    """xyz [B, n, 3] -> [B] int64: pairs i < j closer than ``clash_distance``.  metrics.py:505-512."""
    xyz = np.asarray(xyz, np.float64)
    d = np.linalg.norm(xyz[:, :, None, :] - xyz[:, None, :, :], axis=-1)
    iu = np.triu_indices(xyz.shape[1], k=1)
    return (d[:, iu[0], iu[1]] < clash_distance).sum(axis=1).astype(np.int64)
