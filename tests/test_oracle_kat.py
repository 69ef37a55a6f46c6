"""Pins the oracle on the known answers held by the reference's own tests.

Every test names the reference test (file:line, relative to AG-Peter/encodermap @ 2025-05-23) whose
expected value it reproduces.  CPU only.
"""
import json
import math
import os

import numpy as np
import pytest
import torch
from scipy.spatial.distance import cdist

from oracle import direct_ops as D
from oracle import reference_ops as R

PI = math.pi
DEFAULT_SIG = (4.5, 12, 6, 1, 2, 6)
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def sig_metric(sig, a, b):
    def f(x, y):
        r = np.linalg.norm(x - y)
        return 1 - (1 + (2 ** (a / b) - 1) * (r / sig) ** a) ** (-b / a)
    return f


def periodic_metric(periodicity):
    # tests/test_pairwise_distances.py:66-76
    def f(i, j):
        dx = np.linalg.norm(j - i)
        if dx > periodicity * 0.5:
            dx = dx - periodicity
        if dx <= -periodicity * 0.5:
            dx = dx + periodicity
        return np.abs(dx)
    return f


# ---- tests/test_pairwise_distances.py ------------------------------------------------------
def test_sigmoid_vs_scipy():  # :81-93
    np.random.seed(1)
    points = np.random.random((100, 2)).astype(np.float32) * 360.0
    expect = cdist(points, points, sig_metric(*DEFAULT_SIG[:3]))
    got = R.sigmoid(*DEFAULT_SIG[:3])(R.pairwise_dist(t64(points))).numpy().squeeze()
    np.testing.assert_allclose(got, expect, atol=1e-3)
    got2 = D.sigmoid(D.pairwise_dist_direct(points), *DEFAULT_SIG[:3])
    np.testing.assert_allclose(got2, expect, atol=1e-6)  # scipy metric ran on float32 rows


def test_periodic_distance_literal():  # :98-108
    a = t64([0, 0, 0])
    b = t64([PI / 2, PI, 3 / 2 * PI])
    assert np.array_equal(R.periodic_distance(a, b, 2 * PI).numpy(), np.array([PI / 2, PI, PI / 2]))


def test_periodic_many_points():  # :110-135
    m = periodic_metric(360.0)
    pts = np.vstack([20.0, 40.0, 340.0]).astype(np.float32)
    np.testing.assert_allclose(R.pairwise_dist_periodic(t64(pts), 360.0).numpy(), cdist(pts, pts, m), atol=1e-5)
    np.random.seed(1)
    pts = np.random.random((100, 1)).astype(np.float32) * 360.0
    # the scipy metric runs on float32 rows: compare the float32 run at the test's atol, the float64 oracle at f32 rounding
    np.testing.assert_allclose(R.pairwise_dist_periodic(torch.tensor(pts), 360.0).numpy(), cdist(pts, pts, m), atol=1e-5)
    np.testing.assert_allclose(R.pairwise_dist_periodic(t64(pts), 360.0).numpy(), cdist(pts, pts, m), rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(D.pairwise_dist_periodic_direct(pts, 360.0), cdist(pts, pts, m), rtol=1e-6, atol=1e-5)


def test_two_point_literals():  # :140-168
    pts = np.array([[1 / 8, 1 / 2], [7 / 8, 1 / 2]], dtype=np.float32)
    np.testing.assert_allclose(R.pairwise_dist_periodic(t64(pts), 1).numpy(), [[0, 1 / 4], [1 / 4, 0]], atol=1e-6)
    np.testing.assert_allclose(R.pairwise_dist_periodic(t64(pts), float("inf")).numpy(), [[0, 6 / 8], [6 / 8, 0]], atol=1e-6)
    out = R.pairwise_dist(t64(pts)).numpy()
    assert out.shape == (1, 2, 2)  # leading 1 for 2-D input
    np.testing.assert_allclose(out, [[[0, 6 / 8], [6 / 8, 0]]], atol=1e-6)
    flat = R.pairwise_dist(t64([[0, 0], [1, 0], [0, 1]]), flat=True).numpy()
    np.testing.assert_allclose(flat, [[1, 1, 2 ** 0.5]], atol=1e-6)
    np.testing.assert_allclose(D.pdist_flat(np.array([[[0, 0], [1, 0], [0, 1]]], float)), [[1, 1, 2 ** 0.5]], atol=1e-12)
    rng = np.random.default_rng(0)
    p = rng.normal(size=(10, 3)).astype(np.float32)
    np.testing.assert_allclose(R.pairwise_dist_periodic(t64(p), 10000).numpy().reshape(-1),
                               R.pairwise_dist(t64(p)).numpy().reshape(-1), atol=1e-6)


def test_em1_em2_literals():  # :186-193 ; misc/distances.py:130-137 doctest
    # em1 sigmoid(r, sig, a, b) == em2 sigmoid(sig,a,b)(r)
    assert D.sigmoid(1.5, 5, 12, 2) == R.sigmoid(5, 12, 2)(1.5)
    np.testing.assert_allclose(R.periodic_distance(t64([[1.5]]), t64([[-3.1]])).numpy(), [[1.68318531]], atol=1e-8)


# ---- tests/test_losses.py ------------------------------------------------------------------
def test_sketchmap_cost_non_periodic_vs_scipy():  # :196-229
    rng = np.random.RandomState(0)
    highd = rng.random_sample((256, 51)).astype("float32") * 100
    lowd = rng.random_sample((256, 2)).astype("float32") * 10
    sig_h = cdist(highd, highd, sig_metric(*DEFAULT_SIG[:3]))
    sig_l = cdist(lowd, lowd, sig_metric(*DEFAULT_SIG[3:]))
    cost_scipy = np.mean(np.square(sig_h - sig_l))
    cost_ref = float(R.sigmoid_loss(t64(highd), t64(lowd), float("inf"), DEFAULT_SIG))
    cost_dir, _ = D.sketchmap_fwd_bwd(highd, lowd, float("inf"), DEFAULT_SIG)
    np.testing.assert_allclose(cost_ref, cost_scipy, rtol=1e-6)
    np.testing.assert_allclose(cost_dir, cost_scipy, rtol=1e-6)  # scipy metric ran on float32 rows
    # float32 run of the reference formulation stays within TF's default 1e-6
    cost_f32 = float(R.sigmoid_loss(torch.tensor(highd), torch.tensor(lowd), float("inf"), DEFAULT_SIG))
    np.testing.assert_allclose(cost_f32, cost_scipy, rtol=1e-5)


def test_sketchmap_cost_periodic_vs_numpy_and_scipy():  # :231-280
    rng = np.random.RandomState(1)
    highd = (rng.random_sample((256, 51)).astype("float32") * 2 * np.pi) - np.pi
    lowd = rng.random_sample((256, 2)).astype("float32") * 10
    # the test's numpy restatement
    a = np.expand_dims(highd.astype(np.float64), axis=1)
    b = np.expand_dims(highd.astype(np.float64), axis=0)
    d = np.abs(b - a)
    d = np.minimum(d, 2 * np.pi - d)
    d = d + np.equal(d, 0.0) * 1e-16
    d = np.linalg.norm(d, axis=2)
    sig_h = D.sigmoid(d, *DEFAULT_SIG[:3])
    sig_l = cdist(lowd, lowd, sig_metric(*DEFAULT_SIG[3:]))
    cost_numpy = np.mean(np.square(sig_h - sig_l))
    cost_ref = float(R.sigmoid_loss(t64(highd), t64(lowd), 2 * np.pi, DEFAULT_SIG))
    cost_dir, _ = D.sketchmap_fwd_bwd(highd, lowd, 2 * np.pi, DEFAULT_SIG)
    np.testing.assert_allclose(cost_ref, cost_numpy, rtol=1e-6)
    np.testing.assert_allclose(cost_dir, cost_numpy, rtol=1e-6)

    def wrapped_sig(x, y):  # periodicity_closure_w_sigmoid
        dv = np.abs(x - y)
        dv = np.minimum(dv, 2 * np.pi - dv)
        return D.sigmoid(np.linalg.norm(dv), *DEFAULT_SIG[:3])
    cost_scipy = np.mean(np.square(cdist(highd, highd, wrapped_sig) - sig_l))
    np.testing.assert_allclose(cost_numpy, cost_scipy, atol=1e-3)


def test_loss_closure_zero_nonzero_matrix():  # :283-495, :815-907
    rng = np.random.default_rng(3)
    x = t64(rng.random((64, 6)))
    # center: latent == 1 -> center_cost_scale ; latent == 0 -> 0
    assert float(R.center_cost(torch.ones(64, 2, dtype=torch.float64), 1e-4)) == pytest.approx(1e-4)
    assert float(R.center_cost(torch.zeros(64, 2, dtype=torch.float64), 1e-4)) == 0.0
    # distance: identical (uniform) input rows and constant latent -> 0
    xu = torch.ones(64, 6, dtype=torch.float64) * 0.3
    for per in (float("inf"), 2 * PI):
        assert float(R.distance_cost(xu, torch.zeros(64, 2, dtype=torch.float64), per, DEFAULT_SIG, 500)) == pytest.approx(0.0, abs=1e-12)
        assert float(R.distance_cost(x, t64(rng.random((64, 2))), per, DEFAULT_SIG, 500)) > 0
    # scale None disables a term
    assert float(R.distance_cost(x, t64(rng.random((64, 2))), 2 * PI, DEFAULT_SIG, None)) == 0.0
    # reg == 0 at l2 == 0 ; auto == 0 on identity
    assert float(R.regularization_cost([x], 0.0)) == 0.0
    assert float(R.auto_cost(x, x, 2 * PI)) == 0.0
    assert float(R.auto_cost(x, x + 0.1, float("inf"), "mean_abs", 1)) == pytest.approx(0.1)
    # cartesian_distance_loss(x, x) == 0 : sigmoid_loss with identical high-d and low-d and identical sigmoid params
    s = (4.5, 12, 6, 4.5, 12, 6)
    assert float(R.sigmoid_loss(x, x, float("inf"), s)) == pytest.approx(0.0, abs=1e-15)


def test_soft_start_scale():  # tests/test_losses.py:918-962 ; callbacks/callbacks.py:584-606
    assert R.soft_start_scale(0, (6, 12), 1) == 0.0
    assert R.soft_start_scale(9, (6, 12), 1) == 0.5
    assert R.soft_start_scale(12, (6, 12), 1) == 1.0
    assert R.soft_start_scale(100, (6, 12), 1) == 1.0
    assert R.soft_start_scale(3, (None, None), 2) == 2.0
    assert R.soft_start_scale(3, (6, 12), None) == 0.0


# ---- tests/test_dihedral_to_cartesian.py, tests/test_losses.py (back-mapping) ----------------
def _tables():
    return json.load(open(os.path.join(GOLD, "reference_kat_tables.json")))


def test_straight_tetrahedral_chain_table():  # tests/test_dihedral_to_cartesian.py:186-197
    tab = np.array(_tables()["tetrahedral_chain_7x3"])
    got = R.straight_tetrahedral_chain(bond_lengths=[1, 2, 3, 1, 2, 3])
    np.testing.assert_allclose(got, tab, rtol=1e-6, atol=1e-6)


def test_helix_table_one_way():  # tests/test_dihedral_to_cartesian.py:98-148 (skipped upstream; equals the one-way map)
    T = _tables()
    tab = np.array(T["helix_33x3"])
    phi = (T["helix_phi_deg"] / 180) * PI + PI
    psi = (T["helix_psi_deg"] / 180) * PI + PI
    dih = t64(np.array([[phi, psi, 0.0] * 10] * 2))
    chain = t64(R.straight_tetrahedral_chain(33)).unsqueeze(0).expand(2, -1, -1)
    one_way = R.dihedral_to_cartesian_one_way(dih, chain).numpy()
    np.testing.assert_allclose(one_way[0], tab, atol=2e-5)
    # the current two-way map differs by a rigid motion only: identical intra-frame distances
    two_way = R.dihedrals_to_cartesian(dih, chain)
    np.testing.assert_allclose(D.pdist_flat(two_way.numpy()), D.pdist_flat(one_way), atol=1e-9)


def test_split_indexing_equivalence():  # tests/test_backmapping_em1_em2.py:2116-2157
    rng = np.random.default_rng(5)
    for A in (4, 5, 6, 7, 12, 33, 100, 457):
        cart = np.arange(A)
        dih = np.arange(A - 3)
        s = A // 2
        # reference slicing (encodermap_tf1/backmapping.py:176-181)
        cr, dr = cart[s - 1:], dih[s - 1:]
        cl, dl = cart[s + 1:: -1], dih[s - 2:: -1] if s - 2 >= 0 else dih[:0]
        # oracle's flip form
        assert np.array_equal(cl, cart[: s + 2][::-1])
        assert np.array_equal(dl, dih[: max(s - 1, 0)][::-1])
        # iteration counts models/models.py:666-671
        assert len(dl) == A // 2 - 1 and len(dr) == (A - 3) - (s - 1)


def test_backmap_roundtrip_dihedrals_and_plane():  # tests/test_backmapping_em1_em2.py:643-669 ; tests/test_losses.py:643-703
    rng = np.random.default_rng(7)
    B, A = 5, 24
    angles = t64(rng.normal(2.0, 0.05, (B, A - 2)))
    dih = t64(rng.uniform(-PI, PI, (B, A - 3)))
    lengths = t64(rng.normal(1.4, 0.02, (B, A - 1)))
    plane = R.chain_in_plane(lengths.mean(0, keepdim=True), angles)
    assert int((plane[..., 2] == 0).sum()) == B * A           # z column all zero
    xyz = R.backmap_layer(angles, dih, lengths)
    back = R.dihedrals_mdtraj_convention(xyz)
    diff = R.periodic_distance(back, dih, 2 * PI)
    assert float(diff.abs().max()) < 1e-9
    # bond lengths and angles are what went in
    bl = torch.linalg.norm(xyz[:, 1:] - xyz[:, :-1], dim=-1)
    np.testing.assert_allclose(bl.numpy(), lengths.mean(0, keepdim=True).expand(B, -1).numpy(), atol=1e-9)
    v1 = xyz[:, :-2] - xyz[:, 1:-1]
    v2 = xyz[:, 2:] - xyz[:, 1:-1]
    cosang = (v1 * v2).sum(-1) / torch.linalg.norm(v1, dim=-1) / torch.linalg.norm(v2, dim=-1)
    np.testing.assert_allclose(torch.acos(cosang).numpy(), angles.numpy(), atol=1e-8)


def test_known_angles_fixture():  # tests/data/known_angles.pdb + known_angles_*.npy (featurisation row, section 8f)
    T = _tables()
    xyz = t64(np.array(T["known_angles_xyz_angstrom"]) / 10.0)  # PDB Angstrom -> nm
    d = torch.linalg.norm(xyz[:, 1:] - xyz[:, :-1], dim=-1).numpy()
    np.testing.assert_allclose(d, np.array(T["known_angles_dists"]), atol=2e-4)
    v1 = xyz[:, :-2] - xyz[:, 1:-1]
    v2 = xyz[:, 2:] - xyz[:, 1:-1]
    ang = torch.acos((v1 * v2).sum(-1) / torch.linalg.norm(v1, dim=-1) / torch.linalg.norm(v2, dim=-1))
    np.testing.assert_allclose(np.rad2deg(ang.numpy()), np.array(T["known_angles_angles"]), atol=1e-2)
    dih = np.rad2deg(R.dihedrals_mdtraj_convention(xyz).numpy())
    # SIGNED dihedrals, compared modulo 360 degrees (the fixture's sign convention is pinned, not only the magnitudes)
    dd = dih - np.array(T["known_angles_dihedrals"])
    assert np.abs((dd + 180.0) % 360.0 - 180.0).max() < 1e-2


def test_ala10_backbone_angle_table():  # tests/test_angles.py:72-103 (MolData.angles of tests/data/Ala10_helix.pdb)
    T = _tables()
    xyz = t64(np.array(T["ala10_backbone_xyz_angstrom"])[None] / 10.0)
    v1 = xyz[:, :-2] - xyz[:, 1:-1]
    v2 = xyz[:, 2:] - xyz[:, 1:-1]
    ang = torch.acos((v1 * v2).sum(-1) / torch.linalg.norm(v1, dim=-1) / torch.linalg.norm(v2, dim=-1))
    # the reference's own assertion is np.allclose with its defaults (rtol 1e-5, atol 1e-8); measured here: 2.4e-7 rad
    np.testing.assert_allclose(ang.numpy()[0], np.array(T["ala10_backbone_angles_rad"]), rtol=1e-5, atol=1e-8)


# ---- the two oracle statements agree (values and gradients) ----------------------------------
@pytest.mark.parametrize("periodicity", [float("inf"), 2 * PI])
@pytest.mark.parametrize("sig", [DEFAULT_SIG, (16.0, 12, 6, 1, 2, 6), (0.2, 3, 6, 1, 2, 6)])
def test_direct_gradient_matches_autograd(periodicity, sig):
    rng = np.random.default_rng(11)
    B, Dd, L = 48, 7, 2
    x = rng.uniform(-PI, PI, (B, Dd)) if periodicity < float("inf") else rng.normal(0, 3, (B, Dd))
    z = rng.normal(0, 1, (B, L))
    z[5] = z[9]              # coincident latent points: zero-gradient mask
    x[3] = x[4]              # coincident high-d points
    zt = t64(z).requires_grad_(True)
    loss_ref = R.sigmoid_loss(t64(x), zt, periodicity, sig) * 500
    (g_ref,) = torch.autograd.grad(loss_ref, zt)
    loss_dir, g_dir = D.sketchmap_fwd_bwd(x, z, periodicity, sig, scale=500)
    np.testing.assert_allclose(loss_dir, float(loss_ref), rtol=1e-9)
    np.testing.assert_allclose(g_dir, g_ref.numpy(), rtol=1e-7, atol=1e-12)
    # row-block partials add up to the whole and own their rows' gradient (section 8e)
    l0, g0 = D.sketchmap_fwd_bwd(x, z, periodicity, sig, 500, 0, 20)
    l1, g1 = D.sketchmap_fwd_bwd(x, z, periodicity, sig, 500, 20, B)
    np.testing.assert_allclose(l0 + l1, loss_dir, rtol=1e-12)
    np.testing.assert_allclose(g0 + g1, g_dir, rtol=1e-12, atol=1e-15)
    assert np.all(g0[20:] == 0) and np.all(g1[:20] == 0)


@pytest.mark.parametrize("variant", ["mean_abs", "mean_square", "mean_norm"])
@pytest.mark.parametrize("periodicity", [float("inf"), 2 * PI, 360.0])
def test_periodic_cost_gradient_matches_autograd(variant, periodicity):
    rng = np.random.default_rng(2)
    lim = PI if periodicity == float("inf") else periodicity / 2
    a = rng.uniform(-lim, lim, (32, 9))
    b = rng.uniform(-lim, lim, (32, 9))
    bt = t64(b).requires_grad_(True)
    loss_ref = R.angle_like_cost(t64(a), bt, periodicity, variant, 2.0, 3.0)
    (g_ref,) = torch.autograd.grad(loss_ref, bt)
    loss_dir, g_dir = D.periodic_cost_fwd_bwd(a, b, periodicity, variant, 3.0 / 2.0)
    np.testing.assert_allclose(loss_dir, float(loss_ref), rtol=1e-12)
    np.testing.assert_allclose(g_dir, g_ref.numpy(), rtol=1e-9, atol=1e-15)


def test_pdist_flat_bwd_matches_autograd():
    rng = np.random.default_rng(4)
    xyz = rng.normal(size=(3, 9, 3))
    xt = t64(xyz).requires_grad_(True)
    out = R.pairwise_dist(xt, flat=True)
    gout = rng.normal(size=tuple(out.shape))
    (g_ref,) = torch.autograd.grad((out * t64(gout)).sum(), xt)
    np.testing.assert_allclose(D.pdist_flat(xyz), out.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(D.pdist_flat_bwd(xyz, gout), g_ref.numpy(), atol=1e-10)


def test_adam_statements_agree():
    rng = np.random.default_rng(9)
    w = rng.normal(size=50)
    p = [t64(w).requires_grad_(True)]
    opt = R.KerasAdam(p, learning_rate=1e-3, clipvalue=1.0)
    m = np.zeros(50)
    v = np.zeros(50)
    wn = w.copy()
    for t in range(1, 6):
        g = rng.normal(size=50) * 3
        opt.apply_gradients(p, [t64(g)])
        wn, m, v = D.keras_adam_step(wn, g, m, v, t)
    np.testing.assert_allclose(wn, p[0].detach().numpy(), rtol=1e-12)
    # first step of Adam with clipped unit gradient moves by ~lr
    assert abs(abs(w[0] - D.keras_adam_step(w, np.full(50, 5.0), np.zeros(50), np.zeros(50), 1)[0][0]) - 1e-3) < 1e-8


def test_metric_known_answers():
    """tests/test_autoencoder.py:412-480 of the reference: the literal coordinate sets of ``test_metrics`` -- mean clashes
    1.0 and 2.0 at 0.1 nm (the commented assertions), rmsd(c, c) = 0 within 1e-3, rmsd(c, c2) != 0."""
    from oracle import metrics_ops as M

    c = np.array([[[0.0, 0.0, 0.0], [0.01, 0.0, 0.0], [2.0, 0.0, 0.0]],
                  [[3.0, 0.0, 0.0], [3.01, 0.0, 0.0], [5.0, 0.0, 0.0]]], np.float32)
    c2 = np.array([[[0.0, 0.0, 0.0], [0.01, 0.0, 0.0], [0.02, 0.0, 0.0]],
                   [[1.0, 0.0, 0.0], [0.99, 0.0, 0.0], [5.0, 0.0, 0.0]]], np.float32)
    assert M.clash_counts(c, 0.1).mean() == 1.0
    assert M.clash_counts(c2, 0.1).mean() == 2.0
    np.testing.assert_allclose(M.rmsd_numpy(c, c), [0.0, 0.0], atol=1e-3)
    assert not np.allclose(M.rmsd_numpy(c, c2), [0.0, 0.0], atol=1e-3)
    # a rigid rotation + translation of a frame leaves the (centroid-free) Kabsch distance at zero; a mirror image does not
    rng = np.random.default_rng(0)
    P = rng.normal(size=(30, 3))
    th = 0.7
    Rm = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]])
    w = np.tile(M.WEIGHTS, 10)
    assert M.kabsch_weighted_rmsd(P, P @ Rm.T + 3.0, w) < 1e-6
    assert M.kabsch_weighted_rmsd(P, P * np.array([1.0, 1.0, -1.0]), w) > 1e-2


# ---- all-atom back-mapping ----------------------------------------------------------
def test_rotation_matrix_published_known_answers():
    """``transformations.rotation_matrix`` (third-party, un-vendored; called at encodermap/misc/backmapping.py:1683): the
    known answers of that package's own docstring."""
    import math

    from oracle import allatom_ops as O

    R = O.rotation_matrix(math.pi / 2, [0, 0, 1], [1, 0, 0])
    assert np.allclose(R.dot([0, 0, 0, 1]), [1, -1, 0, 1])
    rng = np.random.default_rng(0)
    angle = (rng.random() - 0.5) * 2 * math.pi
    direc, point = rng.random(3) - 0.5, rng.random(3) - 0.5
    assert np.allclose(O.rotation_matrix(angle, direc, point), O.rotation_matrix(angle - 2 * math.pi, direc, point))
    assert np.allclose(O.rotation_matrix(angle, direc, point), O.rotation_matrix(-angle, -direc, point))
    assert np.allclose(np.identity(4), O.rotation_matrix(2 * math.pi, direc))
    assert np.allclose(2.0, np.trace(O.rotation_matrix(math.pi / 2, direc, point)))


def test_allatom_dihedral_on_the_known_angles_frames():
    """tests/test_backmapping_em1_em2.py:2096-2113 (every dihedral of known_angles lies in (-pi, pi]) and the committed
    known_angles dihedral table (tests/data/known_angles_dihedrals.npy, degrees)."""
    import json
    import os

    from oracle import allatom_ops as O

    with open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat_tables.json")) as f:
        G = json.load(f)
    xyz = np.array(G["known_angles_xyz_angstrom"], np.float64)
    want = np.array(G["known_angles_dihedrals"], np.float64).reshape(-1)
    for frame, w in zip(xyz, want):
        d = O.dihedral(frame, [0, 1, 2, 3])
        assert -np.pi < d <= np.pi
        dd = abs(np.rad2deg(d) - w) % 360
        assert min(dd, 360 - dd) < 0.05


def test_allatom_oracle_realises_the_requested_dihedrals():
    """The reference's ``verify_every_rotation`` check (backmapping.py:1687-1690) on the oracle itself, plus the far-side
    rule of misc/rotate.py:483-497 (the far side holds the bond's second atom, never its first)."""
    from molecules import dihedrals_of, random_molecule
    from oracle import allatom_ops as O

    xyz, bonds, cq, sq = random_molecule(14, 2)
    far_c = [O.far_side(len(xyz), bonds, q[1:3]) for q in cq]
    far_s = [O.far_side(len(xyz), bonds, q[1:3]) for q in sq]
    for q, f in zip(np.concatenate([cq, sq]), far_c + far_s):
        assert q[2] in f and q[3] in f and q[1] not in f and q[0] not in f
    rng = np.random.default_rng(3)
    d = rng.uniform(-np.pi, np.pi, (3, len(cq))).astype(np.float32)
    s = rng.uniform(-np.pi, np.pi, (3, len(sq))).astype(np.float32)
    out = O.allatom_backmap(xyz, cq, far_c, d, sq, far_s, s)
    w = lambda x: (x + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(w(dihedrals_of(out, cq) - d)).max() < 1e-3 and np.abs(w(dihedrals_of(out, sq) - s)).max() < 1e-3
    with pytest.raises(Exception, match="not part of the bond graph"):
        O.far_side(len(xyz), bonds, [cq[0][0], cq[0][3]])
    ring = np.concatenate([bonds, [[cq[0][0], cq[0][3]]]])
    with pytest.raises(Exception, match="components"):
        O.far_side(len(xyz), ring, cq[0][1:3])
