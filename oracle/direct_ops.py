"""Direct-difference float64 NumPy statement of the kernels' math with analytic
gradients -- TEST INFRASTRUCTURE (see oracle/__init__.py).

These are the formulas of SURVEY.md appendix B, i.e. what the HIP kernels
implement; ``oracle.reference_ops`` is the independent reference-formulation
statement they are checked against (tests/test_oracle_kat.py).
"""
from __future__ import annotations

import math

import numpy as np

PI = math.pi


def sigmoid(r, sig, a, b):
    """encodermap/misc/distances.py:86."""
    return 1.0 - (1.0 + (2.0 ** (a / b) - 1.0) * (r / sig) ** a) ** (-b / a)


def sigmoid_prime(r, sig, a, b):
    """d/dr of the above: b*c*(r/sig)^(a-1)/sig * (1 + c (r/sig)^a)^(-b/a - 1)."""
    c = 2.0 ** (a / b) - 1.0
    u = r / sig
    return b * c * u ** (a - 1.0) / sig * (1.0 + c * u ** a) ** (-b / a - 1.0)


def pairwise_dist_direct(x):
    """Euclidean all-pairs distance by direct differences (encodermap/misc/distances.py:179-255 is the
    Gram form of the same quantity; exact 0 on coincident points either way)."""
    x = np.asarray(x, np.float64)
    d2 = np.zeros((x.shape[0], x.shape[0]))
    for k in range(x.shape[1]):
        diff = x[:, None, k] - x[None, :, k]
        d2 += diff * diff
    return np.sqrt(d2)


def pairwise_dist_periodic_direct(x, periodicity):
    """encodermap/misc/distances.py:144-176 without the [B,B,D] temporary."""
    x = np.asarray(x, np.float64)
    s = np.zeros((x.shape[0], x.shape[0]))
    for k in range(x.shape[1]):
        d = np.abs(x[None, :, k] - x[:, None, k])
        v = np.minimum(d, periodicity - d)
        v = v + (v == 0.0) * 1e-12
        s += v * v
    return np.sqrt(s) + 1e-12


def sketchmap_fwd_bwd(x, z, periodicity, sig, scale=1.0, row_begin=0, row_end=None):
    """Sketch-map cost (loss_functions.py:351-367) and its gradient wrt the latent, rows
    [row_begin,row_end) only (SURVEY.md section 8e row-block partition).

    Returns (loss_partial, dZ[B,L] with zeros outside the block) where
    ``loss_partial = scale/B^2 * sum_{i in block} sum_j (S_h(dh_ij)-S_l(dl_ij))^2`` and
    ``dZ[k] = dL/dz_k`` of the FULL loss for k in the block (uses the i<->j symmetry).
    """
    x = np.asarray(x, np.float64)
    z = np.asarray(z, np.float64)
    B = x.shape[0]
    row_end = B if row_end is None else row_end
    if periodicity == float("inf"):
        dh = pairwise_dist_direct(x)
    else:
        dh = pairwise_dist_periodic_direct(x, periodicity)
    dl = pairwise_dist_direct(z)
    sh = sigmoid(dh, *sig[:3])
    sl = sigmoid(dl, *sig[3:])
    diff = sh - sl
    rows = slice(row_begin, row_end)
    loss = scale * np.sum(diff[rows] ** 2) / (B * B)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(dl > 0, -2.0 * scale * diff * sigmoid_prime(dl, *sig[3:]) / (B * B) / dl, 0.0)
    dZ = np.zeros_like(z)
    zi = z[rows]
    # dL/dz_k = 2 * sum_j w_kj (z_k - z_j)
    dZ[rows] = 2.0 * (w[rows].sum(1)[:, None] * zi - w[rows] @ z)
    return loss, dZ


def sketchmap_tiles_fwd_bwd(x, z, periodicity, sig, scale, tiles, tile):
    """The same cost over a SET OF TILES of the symmetric pair matrix (symmetric row-block
    partition): ``tiles`` holds (ti, tj) with ti <= tj, rows [ti*tile, (ti+1)*tile) x columns [tj*tile, (tj+1)*tile).

    An off-diagonal tile stands for its mirror image too: its pairs count twice in the cost and contribute to the latent
    gradient of the rows on BOTH sides.  Returns (loss share, dZ[B,L] share); summed over a partition of the upper
    triangle they are the full-batch cost and gradient of sketchmap_fwd_bwd."""
    x = np.asarray(x, np.float64)
    z = np.asarray(z, np.float64)
    B = x.shape[0]
    dh = pairwise_dist_direct(x) if periodicity == float("inf") else pairwise_dist_periodic_direct(x, periodicity)
    dl = pairwise_dist_direct(z)
    diff = sigmoid(dh, *sig[:3]) - sigmoid(dl, *sig[3:])
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(dl > 0, -2.0 * scale * diff * sigmoid_prime(dl, *sig[3:]) / (B * B) / dl, 0.0)
    loss, dZ = 0.0, np.zeros_like(z)
    for ti, tj in tiles:
        I = slice(ti * tile, min((ti + 1) * tile, B))
        J = slice(tj * tile, min((tj + 1) * tile, B))
        mult = 1.0 if ti == tj else 2.0
        loss += mult * scale * np.sum(diff[I, J] ** 2) / (B * B)
        wt = w[I, J]
        dZ[I] += 2.0 * (wt.sum(1)[:, None] * z[I] - wt @ z[J])
        if ti != tj:
            dZ[J] += 2.0 * (wt.sum(0)[:, None] * z[J] - wt.T @ z[I])
    return loss, dZ


def pdist_flat(xyz):
    """pairwise_dist(flat=True) (encodermap/misc/distances.py:235-253): strict upper triangle, row-major."""
    xyz = np.asarray(xyz, np.float64)
    C = xyz.shape[1]
    iu, ju = np.triu_indices(C, 1)
    diff = xyz[:, iu] - xyz[:, ju]
    return np.sqrt((diff * diff).sum(-1))


def pdist_flat_bwd(xyz, gout):
    """Gradient of pdist_flat wrt xyz; zero contribution where the distance is exactly 0 (reference mask)."""
    xyz = np.asarray(xyz, np.float64)
    C = xyz.shape[1]
    iu, ju = np.triu_indices(C, 1)
    diff = xyz[:, iu] - xyz[:, ju]
    d = np.sqrt((diff * diff).sum(-1))
    with np.errstate(divide="ignore", invalid="ignore"):
        coef = np.where(d > 0, gout / d, 0.0)[..., None] * diff
    g = np.zeros_like(xyz)
    np.add.at(g, (slice(None), iu), coef)
    np.add.at(g, (slice(None), ju), -coef)
    return g


def periodic_in(x, periodicity):
    x = np.asarray(x, np.float64)
    if periodicity != 2 * PI:
        x = x / periodicity * 2 * PI
    return np.concatenate([np.sin(x), np.cos(x)], axis=1)


def periodic_out(y, periodicity):
    y = np.asarray(y, np.float64)
    n = y.shape[1] // 2
    out = np.arctan2(y[:, :n], y[:, n:])
    if periodicity != 2 * PI:
        out = out / (2 * PI) * periodicity
    return out


def periodic_cost_fwd_bwd(a, b, periodicity, variant, coef):
    """coef * variant(periodic_distance(a, b)) and d/db (loss_functions.py:598-613).

    sign convention of TF autodiff: d|b-a|/db = sign(b-a); min picks the first branch on ties.
    """
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    diff = b - a
    d = np.abs(diff)
    if periodicity == float("inf"):
        delta = d
        ddelta = np.sign(diff)
    else:
        wrap = periodicity - d
        delta = np.minimum(d, wrap)
        ddelta = np.where(d <= wrap, 1.0, -1.0) * np.sign(diff)
    n = a.size
    if variant == "mean_abs":
        loss = np.abs(delta).mean()
        g = np.sign(delta) * ddelta / n
    elif variant == "mean_square":
        loss = (delta ** 2).mean()
        g = 2 * delta * ddelta / n
    elif variant == "mean_norm":
        nrm = np.sqrt((delta ** 2).sum(1))
        loss = nrm.mean()
        with np.errstate(divide="ignore", invalid="ignore"):
            g = np.where(nrm[:, None] > 0, delta / nrm[:, None], 0.0) * ddelta / a.shape[0]
    else:
        raise ValueError(variant)
    return coef * loss, coef * g


def keras_adam_step(w, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7, clip=1.0, l2coef=None):
    """One Keras-2.15 Adam update on flat float64 arrays.  ``step`` is t (1-based).
    ``l2coef`` is the per-element 2*lambda (kernels) or 0 (biases): the L2 term's gradient joins the
    data gradient BEFORE the clip, as tape.gradient(loss incl. regulariser) does."""
    g = np.asarray(g, np.float64).copy()
    if l2coef is not None:
        g = g + l2coef * w
    if clip is not None:
        g = np.clip(g, -clip, clip)
    m = m + (g - m) * (1 - b1)
    v = v + (g * g - v) * (1 - b2)
    alpha = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    w = w - alpha * m / (np.sqrt(v) + eps)
    return w, m, v
