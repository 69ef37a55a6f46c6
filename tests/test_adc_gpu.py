"""AngleDihedralCartesianEncoderMap on the GPU against the float64 oracle of ADCFunctionalModel.get_loss /
train_step (oracle.reference_ops.adc_loss, following models/models.py:2260-2521).  Needs an MI355X.
Tolerance: total loss 5e-5, every term 2e-4 relative per step over 6 steps (identical weights and batch indices)."""
import math

import json
import os

import numpy as np
import pytest
import torch

from oracle import reference_ops as R

# total loss / every cost term per step: the north star's 1e-4 halved for the total; measured worst cases 1.3e-5 / 7.6e-5 (the center
# cost of the fourteen-layer network after six free-running steps): profiles/r06_adc_measured_errors.json
TOTAL_RTOL, TERM_RTOL = 5e-5, 2e-4
pytestmark = pytest.mark.gpu
PI = math.pi


@pytest.fixture(scope="module")
def em():
    import encodermap_amd as em
    from encodermap_amd import _lib

    _lib.require_gpu()
    return em


def weights_for(em, p, net_dim, seed=5):
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    lay = build_layout(p.n_neurons, p.activation_functions, net_dim)
    return unpack_weights(lay, init_parameters(lay, seed))


CASES = [
    ("dihedrals_only", dict(), 8, 0, 64),
    ("angles_dihedrals_softstart", dict(use_backbone_angles=True, angle_cost_scale=2, cartesian_cost_scale_soft_start=(2, 5),
                                        cartesian_pwd_start=1, cartesian_pwd_step=3), 10, 0, 48),
    ("sidechains_distance_cost", dict(use_backbone_angles=True, use_sidechains=True, distance_cost_scale=3.0,
                                      dist_sig_parameters=(6.0, 12, 6, 1, 2, 6), cartesian_cost_variant="mean_square",
                                      dihedral_cost_variant="mean_norm", cartesian_dist_sig_parameters=(8.0, 12, 6, 1, 2, 6)), 9, 11, 50),
    ("c3b_shape", dict(cartesian_pwd_start=1, cartesian_pwd_step=3, cartesian_dist_sig_parameters=(20.0, 10, 5, 1, 2, 5),
                       dihedral_cost_reference=0.7, cartesian_cost_reference=3.0), 76, 0, 256),
    # fourteen Dense layers (the reference stacks any number, models/models.py:3133-3160): windows of <= 12 layers in separate launches
    ("fourteen_layers", dict(use_backbone_angles=True, n_neurons=[48, 40, 32, 24, 16, 12, 2],
                             activation_functions=["", "tanh", "tanh", "tanh", "tanh", "tanh", "tanh", ""]), 8, 0, 64),
    # a 410-residue backbone: 1227 dihedrals enter and leave the network 2454 columns wide -- beyond the LDS images of the row-block
    # kernels, so the network's ends run through the streaming kernel (csrc/wide.h); teacher-forced like the 228-atom chain
    ("wide_protein", dict(cartesian_pwd_start=1, cartesian_pwd_step=3, cartesian_dist_sig_parameters=(40.0, 10, 5, 1, 2, 5),
                          dihedral_cost_reference=0.7, cartesian_cost_reference=3.0), 410, 0, 16),
]


@pytest.mark.parametrize("name,kw,n_res,n_side,B", CASES, ids=[c[0] for c in CASES])
def test_adc_loss_trajectory(em, name, kw, n_res, n_side, B):
    rng = np.random.default_rng(n_res)
    data = R.synthetic_backbone(rng, 700, n_res, n_side)
    n_steps = 3 if name == "wide_protein" else 6      # (the float64 oracle of a 1230-atom chain costs ~8 s per step)
    p = em.ADCParameters(n_steps=n_steps, batch_size=B, **kw)
    A = 3 * n_res
    net_dim = 2 * ((A - 2 if p.use_backbone_angles else 0) + (A - 3) + (n_side if p.use_sidechains else 0))
    w0 = weights_for(em, p, net_dim)
    e = em.AngleDihedralCartesianEncoderMap(data, p, model=[w.copy() for w in w0], read_only=True)
    idx = rng.integers(0, 700, (n_steps, B))
    model = R.ADCOracle(A, n_side, p.to_dict(), w0)
    if name not in ("c3b_shape", "wide_protein"):
        e._engine.run_steps(n_steps, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
        got = e._engine.losses().astype(np.float64)
        terms = e._engine.loss_terms()
        want, want_terms = R.adc_train_steps(model, data, idx)
    else:
        # 228-atom chain: the cartesian cost through ~110 chained rotations per direction makes the first Adam steps chaotic
        # (free-running float32 and float64 trajectories agree to 3e-6 for two steps, then separate), so the step function
        # is checked with teacher forcing: the engine starts every step from the oracle's weights and Adam moments.
        from encodermap_amd.layout import pack_weights

        eng = e._engine
        opt = R.KerasAdam(model.params, learning_rate=p.learning_rate, clipvalue=1.0)
        t = {k: torch.tensor(v, dtype=torch.float64) for k, v in data.items()}
        want, want_terms = [], []
        for n in range(n_steps):
            eng.load_state_dict({"w": pack_weights(eng.layout, [q.detach().numpy() for q in model.params]),
                                 "m": pack_weights(eng.layout, [q.numpy() for q in opt.m]),
                                 "v": pack_weights(eng.layout, [q.numpy() for q in opt.v]), "step": n})
            eng.run_steps(1, torch.from_numpy(idx[n]).to("cuda:0"))
            ii = torch.as_tensor(idx[n])
            total, tm = R.adc_loss(model, t["central_angles"][ii], t["central_dihedrals"][ii], t["central_cartesians"][ii],
                                   t["central_distances"][ii], None, n)
            opt.apply_gradients(model.params, torch.autograd.grad(total, model.params))
            want.append(float(total.detach()))
            want_terms.append({k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in tm.items()})
        got = eng.losses().astype(np.float64)
        terms = eng.loss_terms()
        for a, b in zip(eng.get_weights(), model.params):       # weights after the last forced step
            diff = np.abs(a - b.detach().numpy())
            assert (diff > 5e-6).mean() < 0.03 and diff.max() < 2.1e-3
    assert (e._engine.plan.windows is not None) == (name == "fourteen_layers")
    # the MEASURED errors behind the bounds below (profiles/r06_adc_measured_errors.json: EMX_RECORD_ADC_ERRORS=<file> appends a record)
    measured = {"case": name, "steps": n_steps, "total": float(np.max(np.abs(got - np.array(want)) / np.abs(want)))}
    for k in want_terms[0]:
        w_ = np.array([t_[k] for t_ in want_terms], np.float64)
        measured[k] = float(np.max(np.abs(np.asarray(terms[k], np.float64) - w_) / np.maximum(np.abs(w_), 1e-7)))
    print("ADC measured max relative errors:", measured)
    if os.environ.get("EMX_RECORD_ADC_ERRORS"):
        with open(os.environ["EMX_RECORD_ADC_ERRORS"], "a") as fh:
            fh.write(json.dumps(measured) + "\n")
    np.testing.assert_allclose(got, want, rtol=TOTAL_RTOL)
    for k in want_terms[0]:
        np.testing.assert_allclose(terms[k], [t_[k] for t_ in want_terms], rtol=TERM_RTOL, atol=1e-7, err_msg=k)
    # inference on identical weights
    e._engine.set_weights([q.detach().numpy() for q in model.params])
    enc_in = ([data["central_angles"]] if p.use_backbone_angles else []) + [data["central_dihedrals"]] + \
             ([data["side_dihedrals"]] if p.use_sidechains else [])
    z = e.encode([a[:100] for a in enc_in])
    z_o = model.encoder([torch.tensor(a[:100], dtype=torch.float64) for a in enc_in]).detach().numpy()
    np.testing.assert_allclose(z, z_o, atol=2e-4)
    dec = e.decode(z_o.astype(np.float32))
    dec = dec if isinstance(dec, list) else [dec]
    dec_o = model.decoder(torch.tensor(z_o.astype(np.float32), dtype=torch.float64))
    for a, b in zip(dec, dec_o):
        d = np.abs(a - b.detach().numpy())
        assert np.minimum(d, 2 * PI - d).max() < 1e-3
    xyz = e.generate(z_o[:10].astype(np.float32))
    assert xyz.shape == (10, A, 3) and np.isfinite(xyz).all()


def test_adc_facade_train_and_references(em, tmp_path):
    rng = np.random.default_rng(3)
    data = R.synthetic_backbone(rng, 1200, 12)
    p = em.ADCParameters(n_steps=40, batch_size=128, main_path=str(tmp_path), checkpoint_step=20, cartesian_pwd_start=1,
                         cartesian_pwd_step=3, cartesian_cost_scale_soft_start=(10, 20), seed=2)

    class Ensemble:          # MockTrajEnsemble of the reference's tests: an object with a CVs dict
        CVs = data

    e = em.AngleDihedralCartesianEncoderMap(Ensemble(), p)
    # train_for_references (autoencoder.py:1816-1937) with injected batches against its oracle restatement: every batch's
    # three costs, and the references they average to
    nb = 1200 // 128
    idx = rng.integers(0, 1200, (nb, 128))
    got_refs = e._engine.reference_costs(nb, data["central_angles"].mean(0), data["central_dihedrals"].mean(0), data["central_distances"].mean(0),
                                         index_block=torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    want_refs = R.train_for_references(data, idx, p.cartesian_pwd_start, p.cartesian_pwd_stop, p.cartesian_pwd_step)
    for k in ("dihedral_cost", "angle_cost", "cartesian_cost"):
        np.testing.assert_allclose(got_refs[k], want_refs[k], rtol=1e-4, err_msg=k)
    hist = e.train()         # runs train_for_references first (all references were 1)
    assert p.dihedral_cost_reference != 1 and p.angle_cost_reference != 1 and p.cartesian_cost_reference != 1
    # the references the facade stored: means over its own (shuffled) batches = the data-set-wide cost of the mean model, up to
    # the sampling noise of nb batches of 128 frames
    full = R.train_for_references(data, [np.arange(1200)], p.cartesian_pwd_start, p.cartesian_pwd_stop, p.cartesian_pwd_step)
    assert p.dihedral_cost_reference == pytest.approx(full["dihedral_cost"][0], rel=0.03)
    assert p.angle_cost_reference == pytest.approx(full["angle_cost"][0], rel=0.03)
    assert p.cartesian_cost_reference == pytest.approx(full["cartesian_cost"][0], rel=0.03)
    assert len(hist.history["loss"]) == 40 and np.isfinite(hist.history["loss"]).all()
    assert hist.history["cartesian_cost"][5] == 0.0 and hist.history["cartesian_cost"][30] > 0      # soft start (10, 20)
    assert hist.history["loss"][-1] < hist.history["loss"][0]
    assert (tmp_path / "saved_model_20.emx.npz").is_file() and (tmp_path / "saved_model_40.emx.npz").is_file()
    z = e.encode()
    assert z.shape == (1200, 2)
    e2 = em.AngleDihedralCartesianEncoderMap.from_checkpoint(Ensemble(), tmp_path)
    np.testing.assert_array_equal(e2.encode(), z)
    with pytest.raises(Exception, match="side_dihedrals"):
        em.AngleDihedralCartesianEncoderMap(data, em.ADCParameters(use_backbone_angles=True, use_sidechains=True), read_only=True)
    with pytest.raises(Exception, match="Backbone angles are required"):
        em.AngleDihedralCartesianEncoderMap(dict(data, side_dihedrals=np.zeros((1200, 3), np.float32)),
                                            em.ADCParameters(use_sidechains=True), read_only=True)


def test_adc_tracked_metrics(em, tmp_path):
    """p.track_clashes / p.track_RMSD (autoencoder.py:2255-2258): one value per step in the History under the metric's
    name, equal to the numpy restatement of ADCClashMetric / ADCRMSDMetric on that step's batch (input frames vs the
    frames the model back-maps), and the stand-alone metric classes give the same numbers."""
    from oracle import metrics_ops as M

    rng = np.random.default_rng(4)
    data = R.synthetic_backbone(rng, 600, 12)            # 36 atoms: a multiple of 3
    p = em.ADCParameters(n_steps=6, batch_size=64, main_path=str(tmp_path), cartesian_pwd_start=1, cartesian_pwd_step=3,
                         track_clashes=True, track_RMSD=True, seed=5, dihedral_cost_reference=0.5, angle_cost_reference=0.5,
                         cartesian_cost_reference=0.5)
    e = em.AngleDihedralCartesianEncoderMap(data, p, read_only=True)
    eng = e._engine
    idx = rng.integers(0, 600, (6, 64))
    clash, rmsd = [], []
    for t in range(6):
        eng.run_steps(1, torch.from_numpy(idx[t]).to("cuda:0"))
        torch.cuda.synchronize()
        xyz = eng.xyz.cpu().numpy()
        # (the step's input frames: behind the step the engine's buffer already holds the rows of the NEXT batch, gathered beside the
        #  optimizer -- unless EMX_ADC_PREFETCH_HEAD=0)
        cart = np.ascontiguousarray(data["central_cartesians"][idx[t]], dtype=np.float32)
        if not eng._prefetch:
            np.testing.assert_allclose(eng.cart.cpu().numpy(), cart, rtol=0, atol=0)
        clash.append(int(M.clash_counts(xyz, 0.1).sum() // 64))
        rmsd.append(float(M.rmsd_numpy(cart, xyz).mean()))
    got = eng.metric_history()
    assert np.abs(got["ADCClashMetric"] - np.array(clash)).max() <= 1      # a pair within rounding of the threshold
    np.testing.assert_allclose(got["ADCRMSDMetric"], rmsd, rtol=2e-4)
    # stand-alone classes (reference interface)
    mc = em.callbacks.ADCClashMetric(distance_unit="nm", parameters=p)
    mc.update_state(None, [None, None, xyz])
    assert abs(mc.result() - clash[-1]) <= 1
    mr = em.ADCRMSDMetric(parameters=p)
    mr.update_state(cart, xyz)
    np.testing.assert_allclose(mr.result(), M.rmsd_numpy(cart, xyz), rtol=2e-4, atol=1e-6)
    with pytest.raises(Exception, match="distance_unit"):
        em.ADCClashMetric(distance_unit="m")
    # through the facade: the History carries the series
    p2 = em.ADCParameters(n_steps=5, batch_size=64, main_path=str(tmp_path), cartesian_pwd_start=1, cartesian_pwd_step=3, track_clashes=True,
                          track_RMSD=True, seed=5, dihedral_cost_reference=0.5, angle_cost_reference=0.5, cartesian_cost_reference=0.5)
    h = em.AngleDihedralCartesianEncoderMap(data, p2, read_only=True).train()
    assert len(h.history["ADCClashMetric"]) == 5 and len(h.history["ADCRMSDMetric"]) == 5
    assert np.isfinite(h.history["ADCRMSDMetric"]).all() and min(h.history["ADCRMSDMetric"]) > 0


def test_adc_free_running_drift_envelope(em):
    """The 228-atom chain (BASELINE.json configs[2]'s shape) free-running, with the float32 drift of the reference formulation
    itself as the yardstick (tests/test_training_gpu.py::test_long_horizon_drift_envelope): the oracle loop in float64 and
    in float32 on identical weights and batches; at every step, for the total loss and every term,
        |engine - f64| <= K |oracle_f32 - f64| + 2e-4 |f64|,   K = 4 for the total loss, 8 for a single term
    (test_adc_loss_trajectory checks the same steps one by one with teacher forcing).  This trajectory is chaotic -- the float32
    oracle leaves the float64 one by 0.1-10 % within the 10 steps -- so the ratio of two such deviations is a noisy statistic:
    over four seeds the largest single-term ratio was 5.4, 4.1, 1.9 and 1.4, that of the total loss 1.8, 3.2, 1.1 and 1.2
    (profiles/r02_d2c_accuracy.txt, tools/adc_drift_seeds.py), while step for step the back-mapping kernels are 3-10x closer to
    float64 than the float32 oracle (same file, tools/d2c_check.py).  A wrong gradient shows up as a ratio in the hundreds.
    Round 3, the same four seeds on this round's kernels: largest single-term ratio 0.52, 4.13, 1.13, 1.45, total loss 0.4, 3.24,
    1.02, 1.29 -- the bounds (4 and 8) stay what four seeds support with a margin of 1.2-1.9."""
    from tests.test_training_gpu import drift_ratio

    name, kw, n_res, n_side, B = CASES[3]
    rng = np.random.default_rng(n_res)
    data = R.synthetic_backbone(rng, 700, n_res, n_side)
    n_steps = 10
    p = em.ADCParameters(n_steps=n_steps, batch_size=B, **kw)
    A = 3 * n_res
    net_dim = 2 * (A - 3)
    w0 = weights_for(em, p, net_dim)
    idx = rng.integers(0, 700, (n_steps, B))
    e = em.AngleDihedralCartesianEncoderMap(data, p, model=[w.copy() for w in w0], read_only=True)
    e._engine.run_steps(n_steps, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    got, got_terms = e._engine.losses(), e._engine.loss_terms()
    runs = {dt: R.adc_train_steps(R.ADCOracle(A, n_side, p.to_dict(), w0, dtype=dt), data, idx) for dt in (torch.float64, torch.float32)}
    (f64, t64), (f32, t32) = runs[torch.float64], runs[torch.float32]
    ratios = {"loss": drift_ratio(got, f64, f32, rel=2e-4)}
    for k in t64[0]:
        ratios[k] = drift_ratio(got_terms[k], [t[k] for t in t64], [t[k] for t in t32], rel=2e-4)
    print("adc drift ratios", {k: round(v, 2) for k, v in ratios.items()},
          "final rel. deviation engine %.2e oracle-f32 %.2e" % (abs(got[-1] - f64[-1]) / abs(f64[-1]), abs(f32[-1] - f64[-1]) / abs(f64[-1])))
    assert ratios["loss"] <= 4.0 and max(ratios.values()) <= 8.0, ratios


@pytest.mark.parametrize("fold", [True, False])
def test_adc_soft_start_with_and_without_folded_bookkeeping(em, fold, monkeypatch):
    """The head of step t + 1 (batch rows, encoder input, SOFT-START SCALE) is prepared on workgroups of the optimizer launch of
    step t.  With the step's bookkeeping folded into the weight-gradient launch the counter has advanced by then; without
    it (EMX_FOLD_FINISH=0) it has not, and the head must take the scale of step t + 1, not t (callbacks.py:584-606).  Both
    forms against the float64 oracle through a soft start that ramps over the run's steps."""
    from encodermap_amd.engine import SequentialEngine

    monkeypatch.setattr(SequentialEngine, "_fold_finish", fold)
    name, kw, n_res, n_side, B = CASES[1]
    kw = dict(kw, cartesian_cost_scale_soft_start=(1, 6), cartesian_cost_scale=3.0)
    rng = np.random.default_rng(17)
    data = R.synthetic_backbone(rng, 700, n_res, n_side)
    n_steps = 8
    p = em.ADCParameters(n_steps=n_steps, batch_size=B, **kw)
    A = 3 * n_res
    net_dim = 2 * ((A - 2) + (A - 3))
    w0 = weights_for(em, p, net_dim)
    e = em.AngleDihedralCartesianEncoderMap(data, p, model=[w.copy() for w in w0], read_only=True)
    assert e._engine._prefetch and e._engine._late_bookkeeping() == fold
    idx = rng.integers(0, 700, (n_steps, B))
    e._engine.run_steps(n_steps, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    got, terms = e._engine.losses().astype(np.float64), e._engine.loss_terms()
    want, want_terms = R.adc_train_steps(R.ADCOracle(A, n_side, p.to_dict(), w0), data, idx)
    np.testing.assert_allclose(got, want, rtol=2e-4)
    cart = np.array([t["cartesian"] for t in want_terms])
    assert cart[0] == 0 and cart[1] == 0 and cart[2] > 0 and cart[3] > cart[2]     # the ramp starts inside the run
    np.testing.assert_allclose(terms["cartesian"], cart, rtol=5e-4, atol=1e-7)


@pytest.mark.parametrize("case", [1, 2])
def test_adc_step_tail_in_one_launch(em, case, monkeypatch):
    """The ADC step with its last two launches as one (emx_mlp_wgrad_adam_local: weight gradients without slice partials, optimizer,
    scalar epilogue -- and the HEAD of the next ADC step on the launch's spare workgroups, soft-start scale included) against the two
    launches: the head's outputs for the step after the last are the same bits (the same device functions on the same rows), the
    counters agree, the loss trajectories agree to float32 rounding of the gradient sums, and the one launch is deterministic."""
    from encodermap_amd.engine import SequentialEngine

    name, kw, n_res, n_side, B = CASES[case]
    kw = dict(kw, cartesian_cost_scale_soft_start=(2, 7), cartesian_cost_scale=3.0)
    rng = np.random.default_rng(23)
    data = R.synthetic_backbone(rng, 700, n_res, n_side)
    n_steps = 9
    idx = rng.integers(0, 700, (n_steps, B))

    def run(local):
        monkeypatch.setattr(SequentialEngine, "_local_wgrad", local)
        p = em.ADCParameters(n_steps=n_steps, batch_size=B, **kw)
        e = em.AngleDihedralCartesianEncoderMap(data, p, read_only=True)
        eng = e._engine
        torch.manual_seed(3)
        eng.w.copy_(torch.randn_like(eng.w) * 0.05)
        eng._sync_derived_state()
        eng.run_steps(n_steps, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
        loss = eng.losses()
        assert eng._tail_ran == local
        return dict(loss=loss, feat=eng.feat.cpu().numpy(), cart=eng.cart.cpu().numpy(), xin=eng.xin.cpu().numpy(), scale=eng.cart_scale.cpu().numpy(),
                    mean=eng.feat_mean.cpu().numpy(), counters=np.array([int(eng.step), int(eng.cursor), int(eng.hstep)]), w=eng.w.cpu().numpy(),
                    sync=eng.wg_sync.cpu().numpy())

    a, b, a2 = run(True), run(False), run(True)
    for k in ("feat", "cart", "xin", "scale", "mean", "counters"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in ("loss", "w", "feat", "xin"):
        np.testing.assert_array_equal(a[k], a2[k], err_msg=k)
    assert (a["sync"] == 0).all()
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=3e-4)
    assert np.isfinite(a["loss"]).all()


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_adc_frame_halves_in_one_launch(em, case, monkeypatch):
    """emx_adc_frame_fwd_bwd -- the back-mapping with its pair distances, the cartesian cost and the way back, the angle-like costs: one
    workgroup per frame, one launch -- against emx_adc_frame_fwd + emx_adc_frame_bwd: whole training runs bit for bit (losses, every
    term, weights), soft start included."""
    from encodermap_amd.adc_engine import ADCEngine

    name, kw, n_res, n_side, B = CASES[case]
    kw = dict(kw, cartesian_cost_scale_soft_start=(2, 6), cartesian_cost_scale=3.0)
    rng = np.random.default_rng(31 + case)
    data = R.synthetic_backbone(rng, 700, n_res, n_side)
    n_steps = 8
    idx = torch.from_numpy(rng.integers(0, 700, (n_steps, B)).reshape(-1)).to("cuda:0")
    out = []
    for both in (True, False):
        monkeypatch.setattr(ADCEngine, "_frame_both_ok", both)
        p = em.ADCParameters(n_steps=n_steps, batch_size=B, **kw)
        e = em.AngleDihedralCartesianEncoderMap(data, p, read_only=True)
        eng = e._engine
        torch.manual_seed(3)
        eng.w.copy_(torch.randn_like(eng.w) * 0.05)
        eng._sync_derived_state()
        eng.run_steps(n_steps, idx)
        loss, terms = eng.losses(), eng.loss_terms()
        out.append((loss, eng.w.cpu().numpy(), terms))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    for k in out[0][2]:
        np.testing.assert_array_equal(out[0][2][k], out[1][2][k], err_msg=k)
    assert np.isfinite(out[0][0]).all()

