"""ADC-path kernels (column mean, chain_in_plane, dihedrals_to_cartesian, flat pair distances, soft start) against
the float64 oracle (oracle/reference_ops.py follows encodermap_tf1/backmapping.py and misc/distances.py; gradients by
torch autograd, as tf.GradientTape does for the reference).  Needs an MI355X.
Tolerances: coordinates 2e-5 * chain extent (float32 composition over up to ~230 rotations), gradients 1e-4 relative
to the largest entry."""
import math

import numpy as np
import pytest
import torch

from oracle import direct_ops as D
from oracle import reference_ops as R

pytestmark = pytest.mark.gpu
PI = math.pi


@pytest.fixture(scope="module")
def L():
    from encodermap_amd import _lib

    _lib.require_gpu()
    _lib.load()
    return _lib


def dev(a):
    return torch.tensor(np.asarray(a), dtype=torch.float32, device="cuda")


def t64(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


def make_chain(rng, B, A):
    lengths = rng.normal(1.4, 0.05, (B, A - 1)).astype(np.float32)
    angles = rng.normal(2.0, 0.1, (B, A - 2)).astype(np.float32)
    dih = rng.uniform(-PI, PI, (B, A - 3)).astype(np.float32)
    return lengths, angles, dih


def test_colmean(L):
    rng = np.random.default_rng(0)
    for B, n in [(256, 227), (1000, 3), (7, 130), (16, 64), (17, 65), (1, 5), (300, 680)]:
        x = rng.normal(1, 2, (B, n)).astype(np.float32)
        out = torch.empty(n, device="cuda")
        L.colmean(dev(x), out)
        np.testing.assert_allclose(out.cpu().numpy(), x.astype(np.float64).mean(0), rtol=2e-6, atol=2e-6)


# (5, 6: the smallest chains; 65, 66, 129, 132, 133: the steps of a part / of the plane chain end exactly at, one before and one
#  behind a multiple of the 64 lanes the scans run over; 1100: nine steps per lane)
@pytest.mark.parametrize("B,A", [(5, 4), (6, 5), (7, 6), (33, 7), (64, 24), (9, 65), (9, 66), (8, 129), (9, 132), (7, 133), (256, 228), (16, 457),
                                 (3, 1100)])
def test_chain_in_plane_and_backmapping(L, B, A):
    rng = np.random.default_rng(A)
    lengths, angles, dih = make_chain(rng, B, A)
    mean_len = lengths.astype(np.float64).mean(0).astype(np.float32)
    # ---- oracle
    ang_t, dih_t = t64(angles, True), t64(dih, True)
    plane_o = R.chain_in_plane(t64(mean_len)[None, :], ang_t)
    xyz_o = R.dihedrals_to_cartesian(dih_t + PI, plane_o)
    gout = rng.normal(0, 1, (B, A, 3))
    (xyz_o * t64(gout)).sum().backward()
    extent = float(xyz_o.detach().abs().max()) + 1.0
    chain = max(1.0, A / 457)     # float32 rounding of the chained rotations grows with their number
    # ---- kernels
    LEN, ANG, DIH = dev(mean_len), dev(angles), dev(dih)
    plane = torch.full((B, A, 3), float("nan"), device="cuda")
    L.chain_in_plane_fwd(LEN, ANG, plane)
    np.testing.assert_allclose(plane.cpu().numpy(), plane_o.detach().numpy(), atol=2e-5 * extent)
    assert torch.all(plane[..., 2] == 0)                                  # tests/test_losses.py:643-648
    xyz = torch.full((B, A, 3), float("nan"), device="cuda")
    L.dihedrals_to_cartesian_fwd(DIH, PI, plane, xyz)
    np.testing.assert_allclose(xyz.cpu().numpy(), xyz_o.detach().numpy(), atol=3e-5 * extent * chain)
    # round trip: the mdtraj-convention dihedrals of the back-mapped chain are the inputs (test_backmapping_em1_em2.py:643-669)
    back = R.dihedrals_mdtraj_convention(xyz.cpu().double())
    diff = R.periodic_distance(back, t64(dih), 2 * PI)
    assert float(diff.max()) < 2e-3 * chain
    # ---- backward
    G = dev(gout)
    ddih = torch.full((B, A - 3), float("nan"), device="cuda")
    dplane = torch.full((B, A, 3), float("nan"), device="cuda")
    L.dihedrals_to_cartesian_bwd(DIH, PI, plane, G, ddih, dplane)
    dang = torch.full((B, A - 2), float("nan"), device="cuda")
    L.chain_in_plane_bwd(LEN, ANG, dplane, dang)
    gd, ga = dih_t.grad.numpy(), ang_t.grad.numpy()
    np.testing.assert_allclose(ddih.cpu().numpy(), gd, rtol=0, atol=1e-4 * chain * np.abs(gd).max())
    np.testing.assert_allclose(dang.cpu().numpy(), ga, rtol=0, atol=1e-4 * chain * np.abs(ga).max())


def test_backmapping_helix_table(L):
    """33x3 helix table of the reference (tests/test_dihedral_to_cartesian.py:98-148): the two-way map reproduces its
    intra-frame distances (it equals the one-way map up to a rigid motion)."""
    import json
    import os

    T = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat_tables.json")))
    tab = np.array(T["helix_33x3"])
    phi = (T["helix_phi_deg"] / 180) * PI + PI
    psi = (T["helix_psi_deg"] / 180) * PI + PI
    dih = np.array([[phi, psi, 0.0] * 10] * 3, np.float32)
    chain = np.repeat(R.straight_tetrahedral_chain(33)[None], 3, 0)
    out = torch.empty(3, 33, 3, device="cuda")
    L.dihedrals_to_cartesian_fwd(dev(dih), 0.0, dev(chain), out)
    np.testing.assert_allclose(D.pdist_flat(out.cpu().numpy().astype(np.float64))[0], D.pdist_flat(tab[None])[0], atol=2e-4)


# the backward pass spreads an atom's terms over 8 / 4 / 2 / 1 lanes (C <= 128 / 256 / 512 / more) and stages the gradient row
# in LDS when it fits (C <= ~180); 1100 atoms: more atoms than threads
@pytest.mark.parametrize("A,start,step,C,B", [(228, 1, 3, 76, 37), (24, 0, 1, 24, 37), (30, 2, 4, 7, 37), (10, 0, 1, 2, 37), (150, 0, 1, 150, 9),
                                             (200, 0, 1, 200, 5), (457, 0, 1, 457, 3), (1100, 0, 1, 1100, 2)])
def test_pdist_flat(L, A, start, step, C, B):
    rng = np.random.default_rng(C)
    xyz = rng.normal(0, 3, (B, A, 3)).astype(np.float32)
    xyz[:, start + step] = xyz[:, start]           # coincident atoms: exact 0 and zero gradient (distances.py:247-253)
    P = C * (C - 1) // 2
    ldo = (P + 3) // 4 * 4
    out = torch.full((B, ldo), float("nan"), device="cuda")
    X = dev(xyz)
    L.pdist_flat_fwd(X, start, step, C, out)
    sel = xyz[:, start: start + step * C: step].astype(np.float64)
    ref = D.pdist_flat(sel)
    np.testing.assert_allclose(out[:, :P].cpu().numpy(), ref, rtol=2e-6, atol=1e-6)
    assert torch.all(out[:, P:] == 0)
    assert float(out[0, 0]) == 0.0
    gout = rng.normal(0, 1, (B, ldo)).astype(np.float32)
    dx = torch.full((B, A, 3), float("nan"), device="cuda")
    L.pdist_flat_bwd(X, dev(gout), start, step, C, dx, False)
    g_sel = D.pdist_flat_bwd(sel, gout[:, :P].astype(np.float64))
    ref_full = np.zeros((B, A, 3))
    ref_full[:, start: start + step * C: step] = g_sel
    np.testing.assert_allclose(dx.cpu().numpy(), ref_full, rtol=0, atol=2e-5 * np.abs(ref_full).max())
    dx2 = torch.ones(B, A, 3, device="cuda")
    L.pdist_flat_bwd(X, dev(gout), start, step, C, dx2, True)
    np.testing.assert_allclose(dx2.cpu().numpy(), 1 + ref_full, rtol=0, atol=2e-5 * np.abs(ref_full).max())
    # literal of the reference: flat distances of [[0,0],[1,0],[0,1]] are [1, 1, sqrt(2)]  (tests/test_pairwise_distances.py:163-168)
    tri = dev(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32))
    o3 = torch.empty(1, 4, device="cuda")
    L.pdist_flat_fwd(tri, 0, 1, 3, o3)
    np.testing.assert_allclose(o3.cpu().numpy()[0, :3], [1, 1, 2 ** 0.5], rtol=1e-6)


def test_soft_start_and_scaled_cost(L):
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    out = torch.zeros(1, device="cuda")
    for t, want in [(0, 0.0), (9, 0.5), (12, 1.0), (100, 1.0)]:      # tests/test_losses.py:918-962
        step.fill_(t)
        L.soft_start_scale(step, 6, 12, 1.0, out)
        assert float(out) == pytest.approx(want)
    step.fill_(3)
    L.soft_start_scale(step, -1, -1, 2.0, out)                      # (None, None) -> constant scale
    assert float(out) == 2.0
    # cartesian cost = coef * scale(device) * mean|inp - out| with the gradient wrt out, plus an external gradient joined in
    rng = np.random.default_rng(1)
    B, P = 50, 28
    inp, outp = rng.normal(0, 1, (B, P)).astype(np.float32), rng.normal(0, 1, (B, P)).astype(np.float32)
    extra = rng.normal(0, 1, (B, P)).astype(np.float32)
    step.fill_(9)
    L.soft_start_scale(step, 6, 12, 1.0, out)
    loss = torch.zeros(1, device="cuda")
    dY = torch.zeros(B, P, device="cuda")
    ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
    L.periodic_cost_fwd_bwd(dev(outp), 0, 0, False, dev(inp), P, float("inf"), "mean_abs", 1.0 / 3.0, 0, B, loss, dY, ws,
                            coef_scale=out, dpred_add=dev(extra))
    lo, g = D.periodic_cost_fwd_bwd(inp, outp, float("inf"), "mean_abs", 0.5 / 3.0)
    assert float(loss) == pytest.approx(lo, rel=1e-5)
    np.testing.assert_allclose(dY.cpu().numpy(), g + extra, atol=1e-6)
    # forward only (dY = NULL): train_for_references uses this
    L.periodic_cost_fwd_bwd(dev(outp), 0, 0, False, dev(inp), P, float("inf"), "mean_square", 2.0, 0, B, loss, None, ws)
    assert float(loss) == pytest.approx(2.0 * np.mean((inp.astype(np.float64) - outp) ** 2), rel=1e-5)


def test_backbone_featurisation(L):
    """emx_backbone_features against (i) the reference's known_angles fixture (tests/data/known_angles.pdb +
    known_angles_{dists,angles,dihedrals}.npy) and (ii) the round trip through the back-mapping kernels."""
    import json
    import os

    import encodermap_amd as em

    T = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat_tables.json")))
    xyz = np.array(T["known_angles_xyz_angstrom"], np.float32) / 10.0          # PDB Angstrom -> nm
    f = em.featurize_backbone(xyz)
    np.testing.assert_allclose(f["central_distances"], np.array(T["known_angles_dists"]), atol=2e-4)
    np.testing.assert_allclose(np.rad2deg(f["central_angles"]), np.array(T["known_angles_angles"]), atol=1e-2)
    # SIGNED dihedrals, compared modulo 360 degrees (the fixture's sign convention is pinned, not only the magnitudes)
    dd = np.rad2deg(f["central_dihedrals"]) - np.array(T["known_angles_dihedrals"])
    assert np.abs((dd + 180.0) % 360.0 - 180.0).max() < 1e-2
    # the 28 backbone angles of Ala10_helix.pdb (reference tests/test_angles.py:72-103)
    ala = em.featurize_backbone(np.array(T["ala10_backbone_xyz_angstrom"], np.float32)[None] / 10.0)
    np.testing.assert_allclose(ala["central_angles"][0], np.array(T["ala10_backbone_angles_rad"]), atol=2e-5)
    # round trip: features of a back-mapped chain are the inputs of the back-mapping
    rng = np.random.default_rng(2)
    B, A = 40, 60
    lengths, angles, dih = make_chain(rng, B, A)
    ml = lengths.mean(0)
    plane = torch.empty(B, A, 3, device="cuda")
    out = torch.empty(B, A, 3, device="cuda")
    L.chain_in_plane_fwd(dev(ml), dev(angles), plane)
    L.dihedrals_to_cartesian_fwd(dev(dih), PI, plane, out)
    g = em.featurize_backbone(out.cpu().numpy())
    np.testing.assert_allclose(g["central_distances"], np.repeat(ml[None], B, 0), atol=2e-5)
    np.testing.assert_allclose(g["central_angles"], angles, atol=1e-4)
    d = np.abs(g["central_dihedrals"] - dih)
    assert np.minimum(d, 2 * PI - d).max() < 1e-3
    # oracle formula (misc/backmapping.py:344-352)
    ref = R.dihedrals_mdtraj_convention(out.cpu().double()).numpy()
    d = np.abs(g["central_dihedrals"] - ref)
    assert np.minimum(d, 2 * PI - d).max() < 1e-4
    # and the ADC class can be fed from coordinates alone
    e = em.AngleDihedralCartesianEncoderMap.from_cartesians(out.cpu().numpy(), em.ADCParameters(batch_size=16, n_steps=3), read_only=True)
    assert e.encode().shape == (B, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,A", [(2, 3), (64, 228), (256, 30), (5, 456)])
def test_clash_count_and_kabsch_rmsd(B, A):
    """emx_clash_count / emx_kabsch_rmsd (ADCClashMetric, ADCRMSDMetric; callbacks/metrics.py:71-153, 470-581) against
    the numpy restatement: random chains, identical frames (rmsd 0), rotated + shifted copies, mirror images (the
    det < 0 branch), and the reference's literal test frames."""
    from encodermap_amd import _lib as L
    from oracle import metrics_ops as M

    rng = np.random.default_rng(A + B)
    a = np.cumsum(rng.normal(0, 0.15, (B, A, 3)), axis=1).astype(np.float32)
    b = (a + rng.normal(0, 0.05, (B, A, 3))).astype(np.float32)
    th = 1.1
    Rm = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1.0]], np.float32)
    b[0] = a[0]                                         # identical
    b[1 % B] = a[1 % B] @ Rm.T                          # rotated (about the origin)
    if B > 3:
        b[2] = a[2] * np.array([1, 1, -1], np.float32)  # mirror image
        b[3] = b[3] * 50.0                              # very different scale
    if B == 2 and A == 3:
        a = np.array([[[0.0, 0.0, 0.0], [0.01, 0.0, 0.0], [2.0, 0.0, 0.0]], [[3.0, 0.0, 0.0], [3.01, 0.0, 0.0], [5.0, 0.0, 0.0]]], np.float32)
        b = np.array([[[0.0, 0.0, 0.0], [0.01, 0.0, 0.0], [0.02, 0.0, 0.0]], [[1.0, 0.0, 0.0], [0.99, 0.0, 0.0], [5.0, 0.0, 0.0]]], np.float32)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    for translate in (True, False):
        out = torch.zeros(B, device="cuda")
        L.kabsch_rmsd(ta, tb, M.WEIGHTS, translate, out)
        want = M.rmsd_numpy(a, b, translate)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(want).max())))
    same = torch.zeros(B, device="cuda")
    L.kabsch_rmsd(ta, ta, M.WEIGHTS, True, same)
    assert float(same.abs().max()) < 1e-3
    for thr in (0.1, 0.35, 1.0):
        counts = torch.zeros(B, dtype=torch.int64, device="cuda")
        L.clash_count(tb, thr, counts)
        want = M.clash_counts(b, thr)
        # pairs within float32 rounding of the threshold may fall on either side
        d = np.linalg.norm(b[:, :, None, :].astype(np.float64) - b[:, None, :, :], axis=-1)
        iu = np.triu_indices(A, k=1)
        near = (np.abs(d[:, iu[0], iu[1]] - thr) < 1e-5 * max(thr, 1.0)).sum(axis=1)
        assert np.all(np.abs(counts.cpu().numpy() - want) <= near)


@pytest.mark.parametrize("periodicity,soft", [(2 * PI, (6.0, 12.0, 1.0)), (360.0, (-1.0, -1.0, 2.5))])
def test_adc_batch_head_equals_its_four_parts(L, periodicity, soft):
    """emx_adc_batch_head == emx_gather_rows x 2 + emx_periodic_in per block + emx_soft_start_scale + emx_colmean, bit for bit, at a cursor
    position inside the index block, with a ragged number of rows per workgroup."""
    rng = np.random.default_rng(17)
    N, B, A, S = 500, 37, 21, 5
    n_ang, n_dih, n_len = A - 2, A - 3, A - 1
    F = n_ang + n_dih + S + n_len
    feat_data = dev(rng.uniform(-periodicity / 2, periodicity / 2, (N, F)))
    cart_data = dev(rng.normal(0, 1, (N, 3 * A)))
    idx = torch.from_numpy(rng.integers(0, N, 4 * B)).cuda()
    cursor = torch.tensor([2], dtype=torch.int64, device="cuda")
    step = torch.tensor([9], dtype=torch.int64, device="cuda")
    parts = [(0, n_ang, 0), (n_ang, n_dih, 2 * n_ang), (n_ang + n_dih, S, 2 * (n_ang + n_dih))]
    W = 2 * (n_ang + n_dih + S)
    feat, cart, xin = (torch.full((B, k), float("nan"), device="cuda") for k in (F, 3 * A, W))
    sc = torch.zeros(1, device="cuda")
    mean = torch.full((F,), float("nan"), device="cuda")
    L.adc_batch_head(feat_data, cart_data, idx, cursor, B, feat, cart, parts, periodicity, xin, step, *soft, sc, feat_mean=mean)
    feat2, cart2, xin2 = (torch.full((B, k), float("nan"), device="cuda") for k in (F, 3 * A, W))
    sc2 = torch.zeros(1, device="cuda")
    mean2 = torch.full((F,), float("nan"), device="cuda")
    L.gather_rows(feat_data, idx, float("inf"), feat2, None, cursor=cursor, B=B)
    L.gather_rows(cart_data, idx, float("inf"), cart2, None, cursor=cursor, B=B)
    for off, n, col in parts:
        L.periodic_in(feat2[:, off: off + n], periodicity, xin2[:, col: col + 2 * n])
    L.soft_start_scale(step, *soft, sc2)
    L.colmean(feat2, mean2)
    torch.cuda.synchronize()
    assert torch.equal(mean, mean2)                                        # (emx_colmean's order of additions)
    rows = idx[2 * B: 3 * B]
    assert torch.equal(feat, feat_data[rows]) and torch.equal(cart, cart_data[rows])
    assert torch.equal(feat, feat2) and torch.equal(cart, cart2) and torch.equal(xin, xin2) and torch.equal(sc, sc2)
    assert float(sc) == pytest.approx(0.5 if soft[0] >= 0 else 2.5)
    with pytest.raises(L.EmxError, match="outside the feature row"):
        L.adc_batch_head(feat_data, cart_data, idx, cursor, B, feat, cart, [(F - 2, 5, 0)], periodicity, xin, step, *soft, sc)


@pytest.mark.parametrize("B,A,start,step,use_ang", [(256, 228, 1, 3, True), (37, 24, 0, 1, False), (5, 457, 0, 1, True), (9, 4, 0, 1, True),
                                                     (16, 131, 2, 4, False)])
def test_adc_frame_forward_equals_its_stages(L, B, A, start, step, use_ang):
    """emx_adc_frame_fwd (PeriodicOutput + BackMapLayer + both PairwiseDistances blocks in one launch, layers.py:957-986,
    1118-1125, 1252-1267) against the single launches it replaces: every output, value for value."""
    rng = np.random.default_rng(A + B)
    n_ang, n_dih = A - 2, A - 3
    C = len(range(A)[start::step])
    P = C * (C - 1) // 2
    ldo = (P + 3) // 4 * 4
    ldy = 2 * (n_ang + n_dih) + 3
    Y = dev(rng.normal(0, 1, (B, ldy)).astype(np.float32))
    cols_ang, cols_dih = (1, 1 + n_ang), (1 + 2 * n_ang, 1 + 2 * n_ang + n_dih)
    mean_len = dev(rng.normal(0.14, 0.005, (A - 1,)).astype(np.float32))
    mean_ang = dev(rng.normal(2.0, 0.05, (n_ang,)).astype(np.float32))
    cart = dev(rng.normal(0, 1, (B, A, 3)).astype(np.float32))
    nan = lambda *s: torch.full(s, float("nan"), device="cuda")
    # ---- the stages
    ang0, dih0, plane0, xyz0, ip0, op0 = nan(B, n_ang), nan(B, n_dih), nan(B, A, 3), nan(B, A, 3), nan(B, ldo), nan(B, ldo)
    L.periodic_out(Y, n_dih, cols_dih[0], cols_dih[1], 2 * PI, dih0)
    if use_ang:
        L.periodic_out(Y, n_ang, cols_ang[0], cols_ang[1], 2 * PI, ang0)
        L.chain_in_plane_fwd(mean_len, ang0, plane0)
    else:
        from encodermap_amd.adc_engine import _Bcast
        L.chain_in_plane_fwd(mean_len, _Bcast(mean_ang.view(1, -1)), plane0)
    L.dihedrals_to_cartesian_fwd(dih0, PI, plane0, xyz0)
    L.pdist_flat_fwd(cart, start, step, C, ip0)
    L.pdist_flat_fwd(xyz0, start, step, C, op0)
    # ---- one launch
    ang1, dih1, plane1, xyz1, ip1, op1 = nan(B, n_ang), nan(B, n_dih), nan(B, A, 3), nan(B, A, 3), nan(B, ldo), nan(B, ldo)
    L.adc_frame_fwd(Y, 2 * PI, cols_ang if use_ang else None, None if use_ang else mean_ang, ang1 if use_ang else None, cols_dih, dih1, mean_len,
                    PI, plane1, xyz1, cart, start, step, C, ip1, op1)
    torch.cuda.synchronize()
    if A == 457:
        with pytest.raises(L.EmxError) as err:   # (the row of 104 196 pair distances does not fit beside the frame)
            L.adc_frame_fwd(Y, 2 * PI, cols_ang, None, ang1, cols_dih, dih1, mean_len, PI, plane1, xyz1, cart, start, step, C, ip1, op1,
                            split_ws=torch.zeros(L.sketchmap_split_bytes(B, P), dtype=torch.uint8, device="cuda"))
        assert err.value.unsupported
    elif P >= 2:
        # ... and with the split of the input pair rows (the all-pairs kernel's operand planes): emx_sketchmap_split, bit for bit
        sw0 = torch.zeros(L.sketchmap_split_bytes(B, P), dtype=torch.uint8, device="cuda")
        sw1 = torch.full_like(sw0, 0x5A)
        sw0.fill_(0x5A)
        L.sketchmap_split(ip0[:, :P], sw0)
        ip3, op3 = nan(B, ldo), nan(B, ldo)
        L.adc_frame_fwd(Y, 2 * PI, cols_ang if use_ang else None, None if use_ang else mean_ang, ang1 if use_ang else None, cols_dih, dih1,
                        mean_len, PI, plane1, xyz1, cart, start, step, C, ip3, op3, split_ws=sw1)
        torch.cuda.synchronize()
        assert torch.equal(ip0, ip3) and torch.equal(op0, op3)
        assert torch.equal(sw0, sw1), f"split rows differ in {int((sw0 != sw1).sum())} bytes"
    pairs = [("dihedrals", dih0, dih1), ("plane", plane0, plane1), ("xyz", xyz0, xyz1), ("input pairs", ip0, ip1), ("output pairs", op0, op1)]
    if use_ang:
        pairs.append(("angles", ang0, ang1))
    for name, x0, x1 in pairs:
        assert torch.isfinite(x1).all(), name
        assert torch.equal(x0, x1), f"{name}: max difference {float((x0 - x1).abs().max())}"
    # without the input frames' block
    op2 = nan(B, ldo)
    L.adc_frame_fwd(Y, 2 * PI, cols_ang if use_ang else None, None if use_ang else mean_ang, ang1 if use_ang else None, cols_dih, dih1, mean_len,
                    PI, plane1, xyz1, None, start, step, C, None, op2)
    assert torch.equal(op0, op2)


def test_adc_frame_forward_refuses_frames_beyond_lds(L):
    B, A = 2, 1100
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(L.EmxError) as err:
        L.adc_frame_fwd(z(B, 4 * A), 2 * PI, (0, A), None, z(B, A - 2), (2 * A, 3 * A), z(B, A - 3), z(A - 1), PI, z(B, A, 3), z(B, A, 3),
                        z(B, A, 3), 0, 1, A, z(B, A * (A - 1) // 2), z(B, A * (A - 1) // 2))
    assert err.value.unsupported


@pytest.mark.parametrize("B,A,start,step,variant,soft", [(256, 228, 1, 3, "mean_abs", True), (64, 180, 0, 3, "mean_square", False),
                                                         (200, 57, 0, 1, "mean_abs", True)])
def test_adc_frame_backward_equals_its_stages(L, B, A, start, step, variant, soft):
    """emx_adc_frame_bwd (cartesian cost + backward of PairwiseDistances + backward of dihedrals_to_cartesian in one launch,
    models.py:2405-2447, 3392-3393) against the three launches it replaces: the cost's workgroup shares and both gradients, value for
    value."""
    rng = np.random.default_rng(A * B)
    C = len(range(A)[start::step])
    P = C * (C - 1) // 2
    ldo = (P + 3) // 4 * 4
    lengths, angles, dih = make_chain(rng, B, A)
    mean_len = dev(lengths.astype(np.float64).mean(0).astype(np.float32))
    ANG, DIH = dev(angles), dev(dih)
    plane, xyz = torch.empty(B, A, 3, device="cuda"), torch.empty(B, A, 3, device="cuda")
    L.chain_in_plane_fwd(mean_len, ANG, plane)
    L.dihedrals_to_cartesian_fwd(DIH, PI, plane, xyz)
    cart = xyz + 0.05 * torch.randn(B, A, 3, device="cuda")
    inp, out = torch.zeros(B, ldo, device="cuda"), torch.zeros(B, ldo, device="cuda")
    L.pdist_flat_fwd(cart, start, step, C, inp)
    L.pdist_flat_fwd(xyz, start, step, C, out)
    scale = torch.tensor([0.37], device="cuda") if soft else None
    nan = lambda *s: torch.full(s, float("nan"), device="cuda")
    ws0, dpair, dxyz, dd0, dp0 = nan(L.EMX_REDUCE_WS), nan(B, ldo), nan(B, A, 3), nan(B, A - 3), nan(B, A, 3)
    L.periodic_cost_fwd_bwd(out, 0, 0, False, inp, P, float("inf"), variant, 1.0 / 8.0, 0, B, None, dpair, ws0, coef_scale=scale)
    L.pdist_flat_bwd(xyz, dpair, start, step, C, dxyz, False)
    L.dihedrals_to_cartesian_bwd(DIH, PI, plane, dxyz, dd0, dp0)
    ws1, dd1, dp1 = nan(L.EMX_REDUCE_WS), nan(B, A - 3), nan(B, A, 3)
    L.adc_frame_bwd(out, inp, variant, 1.0 / 8.0, scale, ws1, xyz, DIH, PI, plane, start, step, C, dd1, dp1)
    torch.cuda.synchronize()
    assert torch.isfinite(ws1).all() and torch.isfinite(dd1).all() and torch.isfinite(dp1).all()
    assert float(ws1.sum()) > 0
    assert torch.equal(ws0, ws1), f"cost shares: max difference {float((ws0 - ws1).abs().max())}"
    assert torch.equal(dd0, dd1), f"d dihedrals: max difference {float((dd0 - dd1).abs().max())}"
    assert torch.equal(dp0, dp1), f"d plane: max difference {float((dp0 - dp1).abs().max())}"
    # ---- with chain_in_plane's backward pass behind it (models with backbone angles)
    da0, da3, dd3, dp3, ws3 = nan(B, A - 2), nan(B, A - 2), nan(B, A - 3), nan(B, A, 3), nan(L.EMX_REDUCE_WS)
    L.chain_in_plane_bwd(mean_len, ANG, dp0, da0)
    L.adc_frame_bwd(out, inp, variant, 1.0 / 8.0, scale, ws3, xyz, DIH, PI, plane, start, step, C, dd3, dp3, mean_lengths=mean_len, angles=ANG,
                    dangles=da3)
    torch.cuda.synchronize()
    assert torch.equal(ws0, ws3) and torch.equal(dd0, dd3) and torch.equal(dp0, dp3)
    assert torch.equal(da0, da3), f"d angles: max difference {float((da0 - da3).abs().max())}"
    # ---- with the angle-like costs of the decoder output behind it in the same launch (emx_periodic_cost_batch's jobs; the dihedral
    #      cost adds d cost / d dihedrals of the cartesian path to its own gradient)
    n_dih, n_ang = A - 3, A - 2
    if 128 <= n_dih <= 512:
        from encodermap_amd.adc_engine import _Bcast
        g = torch.Generator().manual_seed(A)
        Y = torch.randn(B, 2 * n_dih + 3, generator=g).cuda()
        feat = ((torch.rand(B, n_dih + n_ang, generator=g) - 0.5) * 6.0).cuda()
        mean_ang = torch.full((1, n_ang), 2.0, device="cuda")

        def jobs(dY, w1, w2, add):
            return [dict(y=Y, off_sin=1, off_cos=1 + n_dih, periodic_output=True, target=feat[:, :n_dih], n=n_dih, periodicity=2 * PI, variant=variant,
                         coef=0.7, dY=dY, ws=w1, dpred_add=add),
                    dict(y=_Bcast(mean_ang), off_sin=0, off_cos=0, periodic_output=False, target=feat[:, n_dih:], n=n_ang, periodicity=2 * PI,
                         variant="mean_abs", coef=0.3, dY=None, ws=w2)]
        dY0, wa0, wb0 = nan(B, Y.shape[1]), nan(L.EMX_REDUCE_WS), nan(L.EMX_REDUCE_WS)
        L.periodic_cost_batch(jobs(dY0, wa0, wb0, dd0), B, 0, B)
        dY1, wa1, wb1, ws2, dd2, dp2 = nan(B, Y.shape[1]), nan(L.EMX_REDUCE_WS), nan(L.EMX_REDUCE_WS), nan(L.EMX_REDUCE_WS), nan(B, A - 3), nan(B, A, 3)
        L.adc_frame_bwd(out, inp, variant, 1.0 / 8.0, scale, ws2, xyz, DIH, PI, plane, start, step, C, dd2, dp2, jobs=jobs(dY1, wa1, wb1, dd2))
        torch.cuda.synchronize()
        assert torch.equal(ws0, ws2) and torch.equal(dd0, dd2) and torch.equal(dp0, dp2)
        cols = torch.zeros(Y.shape[1], dtype=torch.bool, device="cuda")
        cols[1: 1 + 2 * n_dih] = True
        assert torch.equal(dY0[:, cols], dY1[:, cols]), f"d cost / d output: max difference {float((dY0[:, cols] - dY1[:, cols]).abs().max())}"
        assert torch.isnan(dY1[:, ~cols]).all()                               # (columns outside the job's windows are not touched)
        assert torch.equal(wa0, wa1) and torch.equal(wb0, wb1)


def test_adc_frame_backward_leaves_other_shapes_to_the_stages(L):
    z = lambda *s: torch.zeros(*s, device="cuda")
    B, A = 16, 30          # 435 pair distances per frame: the stand-alone cost launch partitions such rows differently
    P = A * (A - 1) // 2
    for variant, b in (("mean_abs", B), ("mean_norm", 16)):
        with pytest.raises(L.EmxError) as err:
            L.adc_frame_bwd(z(b, P + 1), z(b, P + 1), variant, 1.0, None, z(L.EMX_REDUCE_WS), z(b, A, 3), z(b, A - 3), PI, z(b, A, 3), 0, 1, A,
                            z(b, A - 3), z(b, A, 3))
        assert err.value.unsupported
    # a cost job of fewer than 128 columns runs a wave per row in the batch launch: not this pass
    b, A2 = 8, 228
    C2 = 76
    P2 = C2 * (C2 - 1) // 2
    with pytest.raises(L.EmxError) as err:
        L.adc_frame_bwd(z(b, P2), z(b, P2), "mean_abs", 1.0, None, z(L.EMX_REDUCE_WS), z(b, A2, 3), z(b, A2 - 3), PI, z(b, A2, 3), 1, 3, C2,
                        z(b, A2 - 3), z(b, A2, 3),
                        jobs=[dict(y=z(b, 80), off_sin=0, off_cos=40, periodic_output=True, target=z(b, 40), n=40, periodicity=2 * PI,
                                   variant="mean_abs", coef=1.0, dY=z(b, 80), ws=z(L.EMX_REDUCE_WS))])
    assert err.value.unsupported


@pytest.mark.parametrize("B,with_means", [(37, True), (256, True), (64, False)])
def test_adc_head_beside_the_optimizer(L, B, with_means):
    """emx_next_batch::adc_head: the optimizer launch does the head of the NEXT ADC step (emx_adc_batch_head for batch cursor + 1) on
    workgroups of its own -- rows, encoder input, soft-start scale, batch means: bit for bit; nothing is touched behind the last
    batch of the index block, and the optimizer's own result does not depend on it."""
    rng = np.random.default_rng(B)
    N, A, S = 500, 21, 5
    n_ang, n_dih, n_len = A - 2, A - 3, A - 1
    F = n_ang + n_dih + S + n_len
    feat_data = dev(rng.uniform(-3, 3, (N, F)))
    cart_data = dev(rng.normal(0, 1, (N, 3 * A)))
    n_batches = 4
    idx = torch.from_numpy(rng.integers(0, N, n_batches * B)).cuda()
    parts = [(0, n_ang, 0), (n_ang, n_dih, 2 * n_ang), (n_ang + n_dih, S, 2 * (n_ang + n_dih))]
    W = 2 * (n_ang + n_dih + S)
    soft = (6.0, 12.0, 1.0)
    n = 3000
    grad = torch.randn(n, device="cuda")
    for cur in (1, n_batches - 1):
        step = torch.tensor([9], dtype=torch.int64, device="cuda")
        cursor = torch.tensor([cur], dtype=torch.int64, device="cuda")
        feat, cart, xin = (torch.full((B, k), -7.0, device="cuda") for k in (F, 3 * A, W))
        sc, mean = torch.full((1,), -7.0, device="cuda"), torch.full((F,), -7.0, device="cuda")
        rec = L.adc_head_describe(feat_data, cart_data, B, feat, cart, parts, 2 * PI, xin, step, *soft, sc, mean if with_means else None, n_batches)
        nxt = L.make_next_adc_head(rec, idx, cursor, B, F if with_means else 0)
        w1, m1, v1 = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        w2, m2, v2 = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        ws = torch.zeros(L.EMX_REDUCE_WS, device="cuda")
        L.adam_step(w1, grad, m1, v1, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, None, ws, next_batch=nxt)
        L.adam_step(w2, grad, m2, v2, None, 1e-2, 0.9, 0.999, 1e-7, 1.0, step, None, ws)
        torch.cuda.synchronize()
        assert torch.equal(w1, w2) and torch.equal(m1, m2) and torch.equal(v1, v2)
        if cur + 1 < n_batches:
            feat2, cart2, xin2 = (torch.full((B, k), -7.0, device="cuda") for k in (F, 3 * A, W))
            sc2, mean2 = torch.full((1,), -7.0, device="cuda"), torch.full((F,), -7.0, device="cuda")
            # (this optimizer call runs BEFORE the step's bookkeeping -- no bookkeeping_done --, so the head it prepares belongs to the
            #  step behind the counter's: batch cursor + 1 AND the soft-start scale of step + 1)
            L.adc_batch_head(feat_data, cart_data, idx, torch.tensor([cur + 1], device="cuda"), B, feat2, cart2, parts, 2 * PI, xin2, step + 1, *soft, sc2,
                             feat_mean=mean2 if with_means else None)
            torch.cuda.synchronize()
            assert torch.equal(feat, feat2) and torch.equal(cart, cart2) and torch.equal(xin, xin2) and torch.equal(sc, sc2)
            assert torch.equal(mean, mean2)
            assert torch.equal(feat, feat_data[idx[(cur + 1) * B: (cur + 2) * B]])
        else:
            assert float(feat.max()) == -7.0 and float(xin.max()) == -7.0 and float(mean.max()) == -7.0   # nothing behind the last batch
