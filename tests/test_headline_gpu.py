"""The BASELINE.json configurations at (or next to) their full sizes, as training STEPS through the engine -- round 3.

* C2 (1e6 x 128 blobs, B = 4096, sig (16, 12, 6, 1, 2, 6)): three training steps against the float64 oracle loop
  (R.train_steps, the reference formulation with its Gram-matrix temporaries: 134 MB each at this size).
* C3a (the cube example, encodermap/examples/cube.py:25-38): EncoderMap on random_on_cube_edges(50000, 0.05), B = 256,
  dist_sig_parameters (0.2, 3, 6, 1, 2, 6) -- the general-exponent sigmoid path -- 12 steps against the oracle.
* C4 (B = 8192, D = 512 periodic) and C5 (B = 16384, D = 256 Euclidean) full-size steps: size-independent properties --
  finite, decreasing, bitwise reproducible -- and, at B = 8192 / 16384, row-block data parallelism with two ranks on the
  one GPU against the single process.
* a new training set through set_data() after a captured run (the captured step holds the old set's pointer).
Needs an MI355X.
"""
import math
import os
import socket

import numpy as np
import pytest
import torch

from oracle import reference_ops as R

pytestmark = pytest.mark.gpu
INF = float("inf")


@pytest.fixture(scope="module")
def em():
    import encodermap_amd as em
    from encodermap_amd import _lib

    _lib.require_gpu()
    return em


def _weights(p, net_dim, seed=3):
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    lay = build_layout(p.n_neurons, p.activation_functions, net_dim)
    return unpack_weights(lay, init_parameters(lay, seed))


def blobs(rng, n, d, centres=16):
    """blob centres normal(0, 4), points centre[k] + normal(0, 1)"""
    c = rng.normal(0, 4, (centres, d))
    return (c[rng.integers(0, centres, n)] + rng.normal(0, 1, (n, d))).astype(np.float32)


def test_c2_three_steps_against_the_float64_oracle(em):
    """BASELINE config 2 at its batch size and width.  Tolerances: total loss 1e-4 relative (the north star's bound), every
    cost term 2e-4."""
    rng = np.random.default_rng(1)
    data = blobs(rng, 20000, 128)
    p = em.Parameters(periodicity=INF, batch_size=4096, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), n_steps=3)
    w0 = _weights(p, 128)
    idx = rng.integers(0, data.shape[0], (3, 4096))
    e = em.EncoderMap(p, data, model=[w.copy() for w in w0], read_only=True)
    e._engine.run_steps(3, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    got, terms = e._engine.losses().astype(np.float64), e._engine.loss_terms()
    model = R.SequentialOracle(128, p.n_neurons, p.activation_functions, p.periodicity, w0)
    want, want_terms = R.train_steps(model, data, idx, p.to_dict())
    print("C2 steps: engine", got, "oracle", np.array(want), "rel", np.abs(got - want) / np.abs(want))
    np.testing.assert_allclose(got, want, rtol=1e-4)
    for k in ("auto", "reg", "center", "distance"):
        np.testing.assert_allclose(terms[k], [t[k] for t in want_terms], rtol=2e-4, atol=1e-9)
    assert want_terms[0]["distance"] > 0   # (the sigmoid is not saturated at these parameters)


@pytest.mark.parametrize("B,n_neurons,variant", [(2048, [128, 128, 2], "mean_abs"), (4000, [128, 128, 3], "mean_square")])
def test_transposed_chain_trajectory_against_the_float64_oracle(em, B, n_neurons, variant):
    """Twelve free-running steps of the kernel C2 actually runs -- the transposed, register-resident chain (csrc/tchain.h) -- against the
    float64 oracle: a batch of half the C2 size, and a RAGGED one (4000 rows = 250 row blocks, the last all-pairs tile column half
    empty) with a three-column latent and the mean-square auto cost.  Tolerances as for C2: total loss 1e-4 relative, cost terms 2e-4."""
    from encodermap_amd import _lib

    rng = np.random.default_rng(B)
    data = blobs(rng, 20000, 128)
    p = em.Parameters(periodicity=INF, batch_size=B, n_neurons=n_neurons, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), n_steps=12,
                      auto_cost_variant=variant)
    w0 = _weights(p, 128)
    idx = rng.integers(0, data.shape[0], (12, B))
    e = em.EncoderMap(p, data, model=[w.copy() for w in w0], read_only=True)
    c0 = _lib.load().emx_transposed_chain_count()
    e._engine.run_steps(12, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    got, terms = e._engine.losses().astype(np.float64), e._engine.loss_terms()
    assert _lib.load().emx_transposed_chain_count() > c0, "the transposed chain did not take these launches"
    model = R.SequentialOracle(128, p.n_neurons, p.activation_functions, p.periodicity, w0)
    want, want_terms = R.train_steps(model, data, idx, p.to_dict())
    print(f"B={B}: engine", got, "oracle", np.array(want), "rel", np.abs(got - want) / np.abs(want))
    np.testing.assert_allclose(got, want, rtol=1e-4)
    for k in ("auto", "reg", "center", "distance"):
        np.testing.assert_allclose(terms[k], [t[k] for t in want_terms], rtol=2e-4, atol=1e-9)


def test_c3a_cube_example(em):
    """encodermap/examples/cube.py:25-38: points on the edges of a cube, Euclidean, sig_h = (0.2, 3, 6) -- a_h = 3 is NOT the
    fast (12, 6) family: the exp2/log2 form of the high-d sigmoid runs."""
    from encodermap_amd.misc import random_on_cube_edges

    data, ids = random_on_cube_edges(50000, sigma=0.05, rng=np.random.default_rng(2))
    data = np.asarray(data, np.float32)
    assert data.shape == (50000, 3)
    rng = np.random.default_rng(2)
    p = em.Parameters(periodicity=INF, batch_size=256, dist_sig_parameters=(0.2, 3, 6, 1, 2, 6), n_steps=12)
    w0 = _weights(p, 3)
    idx = rng.integers(0, data.shape[0], (12, 256))
    e = em.EncoderMap(p, data, model=[w.copy() for w in w0], read_only=True)
    e._engine.run_steps(12, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    got, terms = e._engine.losses().astype(np.float64), e._engine.loss_terms()
    model = R.SequentialOracle(3, p.n_neurons, p.activation_functions, p.periodicity, w0)
    want, want_terms = R.train_steps(model, data, idx, p.to_dict())
    np.testing.assert_allclose(got, want, rtol=1e-4)
    for k in ("auto", "reg", "center", "distance"):
        np.testing.assert_allclose(terms[k], [t[k] for t in want_terms], rtol=2e-4, atol=1e-9)
    e._engine.set_weights([b.detach().numpy() for b in model.params])
    z = e.encode(data[:500])
    z_o = model.encoder(torch.tensor(data[:500], dtype=torch.float64)).detach().numpy()
    np.testing.assert_allclose(z, z_o, atol=1e-4)


FULL = {
    # name: (parameters, data maker) -- BASELINE configs 4 and 5 at their batch size and width (the data sets shortened: the
    # step does not depend on N)
    "c4": (dict(batch_size=8192, dist_sig_parameters=(30.0, 12, 6, 1, 2, 6)),
           lambda r: r.uniform(-np.pi, np.pi, (40000, 512)).astype(np.float32)),
    "c5": (dict(periodicity=INF, batch_size=16384, dist_sig_parameters=(22.6, 12, 6, 1, 2, 6)), lambda r: blobs(r, 60000, 256)),
}


def _full_run(em, name, n_steps, dp=None):
    kw, maker = FULL[name]
    data = maker(np.random.default_rng(4))
    p = em.Parameters(n_steps=n_steps, seed=9, **kw)
    D = data.shape[1]
    w0 = _weights(p, 2 * D if p.periodicity < INF else D)
    idx = np.random.default_rng(6).integers(0, data.shape[0], (n_steps, p.batch_size))
    kwargs = dict(dp=dp) if dp is not None else {}
    e = em.EncoderMap(p, data, model=w0, read_only=True, **kwargs)
    e._engine.run_steps(n_steps, torch.from_numpy(idx.reshape(-1)).to("cuda:0"))
    return e


@pytest.mark.parametrize("name", ["c4", "c5"])
def test_full_size_steps_properties(em, name):
    """20 full-size steps: finite, the cost goes down, and a second run reproduces the first bit for bit (no atomics in any
    reduction of the step: fixed-order partial sums everywhere)."""
    runs = []
    for _ in range(2):
        e = _full_run(em, name, 20)
        runs.append((e._engine.losses(), e._engine.w.cpu().numpy(), {k: v.copy() for k, v in e._engine.loss_terms().items()}))
        del e
        torch.cuda.empty_cache()
    loss, w, terms = runs[0]
    assert np.isfinite(loss).all() and np.isfinite(w).all()
    assert loss[-1] < loss[0], loss
    assert all(np.isfinite(v).all() for v in terms.values()) and terms["distance"][0] > 0
    np.testing.assert_array_equal(loss, runs[1][0])
    np.testing.assert_array_equal(w, runs[1][1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rowblock_worker(rank, world, port, name, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import encodermap_amd as em
    from encodermap_amd.engine import DataParallel

    e = _full_run(em, name, 4, dp=DataParallel(rank, world, "rowblock"))
    np.savez(os.path.join(out_dir, f"{name}_{rank}.npz"), loss=e._engine.losses(), w=e._engine.w.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["c4", "c5"])
def test_full_size_rowblock_two_ranks_equal_single_process(em, name, tmp_path):
    """The objective-preserving partition at the full batch: two row-block ranks on the one GPU (gloo
    collectives), dealt all-pairs tiles, against the single process: losses to 2e-5 over four steps."""
    import torch.multiprocessing as mp

    mp.spawn(_rowblock_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    runs = [np.load(tmp_path / f"{name}_{r}.npz") for r in range(2)]
    np.testing.assert_array_equal(runs[0]["w"], runs[1]["w"])
    np.testing.assert_array_equal(runs[0]["loss"], runs[1]["loss"])
    e = _full_run(em, name, 4)
    np.testing.assert_allclose(runs[0]["loss"], e._engine.losses(), rtol=2e-5)
    diff = np.abs(runs[0]["w"] - e._engine.w.cpu().numpy())
    assert (diff > 2e-5).mean() < 0.02 and diff.max() < 2e-3, (diff.max(), (diff > 2e-5).mean())


@pytest.mark.parametrize("periodic", [True, False])
def test_new_training_set_after_a_captured_run(em, periodic):
    """Train, hand the engine ANOTHER training set (set_data / set_train_data), train again: the captured step held the old
    set's pointer and row count, so the graph must be dropped -- the second leg must equal a fresh engine that starts from
    the first leg's state on the new set, bit for bit."""
    rng = np.random.default_rng(8)
    mk = (lambda n: rng.uniform(-math.pi, math.pi, (n, 10)).astype(np.float32)) if periodic else (lambda n: blobs(rng, n, 32, 6))
    kw = dict() if periodic else dict(periodicity=INF, dist_sig_parameters=(8.0, 12, 6, 1, 2, 6))
    a, b = mk(3000), mk(1700)          # (a different row count too)
    p = em.Parameters(n_steps=20, batch_size=256, **kw)
    w0 = _weights(p, 20 if periodic else 32)
    ia = rng.integers(0, 3000, (10, 256))
    ib = rng.integers(0, 1700, (10, 256))
    e = em.EncoderMap(p, a, model=[w.copy() for w in w0], read_only=True)
    e._engine.run_steps(10, torch.from_numpy(ia.reshape(-1)).cuda())
    state = e._engine.state_dict()
    del a
    e._engine.set_data(b)
    e._engine.run_steps(10, torch.from_numpy(ib.reshape(-1)).cuda())
    second = e._engine.losses(10)
    f = em.EncoderMap(p, b, model=[w.copy() for w in w0], read_only=True)
    f._engine.load_state_dict(state)
    f._engine.run_steps(10, torch.from_numpy(ib.reshape(-1)).cuda())
    np.testing.assert_array_equal(second, f._engine.losses())
    np.testing.assert_array_equal(e._engine.w.cpu().numpy(), f._engine.w.cpu().numpy())


def test_shuffle_stream_blocks_do_not_copy_the_pass(em):
    """draw_index_block: consecutive draws inside one pass are views of ONE permutation (no copy + clone of the remainder
    per call), a draw that straddles passes continues exactly where the host twin (layout.BatchIndexStream on the same
    keyed bijection) does."""
    from encodermap_amd.layout import feistel_permutation

    rng = np.random.default_rng(3)
    data = rng.uniform(-3, 3, (1000, 4)).astype(np.float32)
    p = em.Parameters(n_steps=5, batch_size=64, seed=7)
    e = em.EncoderMap(p, data, read_only=True)
    eng = e._engine
    seed = eng._shuffle_seed
    blocks = [eng.draw_index_block(k).cpu().numpy().copy() for k in (3, 5, 9, 20, 1)]     # 38 batches of 64 = 2432 rows: 3 passes
    got = np.concatenate(blocks)
    want = np.concatenate([feistel_permutation(1000, seed, ep) for ep in range(3)])[: got.size]
    np.testing.assert_array_equal(got, want)
    base = eng._perm.data_ptr()
    v = eng.draw_index_block(2)
    assert v.data_ptr() >= base and v.data_ptr() < base + eng._perm.numel() * 8      # a view into the current pass


OVERLAP_CASES = [
    # B, D, steps, sig, periodicity, data span in periods (periodic data only)
    (4096, 128, 40, None, INF, 0), (1000, 128, 24, None, INF, 0), (640, 32, 24, None, INF, 0), (2048, 256, 16, None, INF, 0), (1024, 64, 16, None, INF, 0),
    (256, 3, 24, (1.5, 3, 6, 1, 2, 6), INF, 0), (512, 100, 16, (12.0, 10, 5, 1.5, 4, 2), INF, 0),      # general sigmoid exponents
    # periodic batches (the default configuration: B = 256, period 2 pi): 32 x 32 VALU tiles, phase form and -- data over more than
    # 1.5 periods -- float form; an odd number of tiles, a ragged last tile, the general sigmoid
    (256, 10, 24, (4.5, 12, 6, 1, 2, 6), 2 * math.pi, 1.0), (512, 6, 16, (90.0, 10, 5, 1.5, 4, 2), 360.0, 1.0), (100, 33, 16, (6.0, 12, 6, 1, 2, 6), 2 * math.pi, 1.0),
    (200, 12, 16, (4.5, 12, 6, 1, 2, 6), 2 * math.pi, 2.0),
]


@pytest.mark.parametrize("B,D,steps,sig,periodicity,span", OVERLAP_CASES)
def test_tiles_inside_the_fused_launch_change_nothing(em, B, D, steps, sig, periodicity, span, monkeypatch):
    """emx_mlp_forward_backward_tiles: the all-pairs tiles as workgroups of the fused dense-layer launch (chain workgroups publish
    the latent rows, tile workgroups wait for them inside the launch) against the separate launches -- the same arithmetic in
    the same order into the same slots, so whole training runs must agree BIT FOR BIT (a stale or torn latent row anywhere in
    `steps` steps x B^2 pairs would show); batches that are not whole tiles / whole groups of eight row blocks included."""
    from encodermap_amd.engine import SequentialEngine

    rng = np.random.default_rng(B + D)
    if periodicity < INF:
        data = rng.uniform(-0.5 * span * periodicity, 0.5 * span * periodicity, (6000, D)).astype(np.float32)
    else:
        data = blobs(rng, 6000, D)
    idx = rng.integers(0, 6000, (steps, B))
    out = []
    monkeypatch.setattr(SequentialEngine, "_overlap_max_rows", 4096)    # (the engine's own limit is 1536 rows: beyond that the separate launches are faster)
    # (bit for bit is a statement about ONE kernel: small non-periodic networks take the transposed chain for the plain fused launch
    #  -- another order of summation; the row-block kernel everywhere here, the default's agreement to rounding below)
    monkeypatch.setenv("EMX_TCHAIN", "0")
    # (overlap, join): the tiles aboard AND the encoder's backward units in the same launch (emx_mlp_forward_backward_tiles_join: the
    # dense-layer workgroups wait inside the launch for the tiles' partial sums) / the tiles aboard / the separate launches
    for overlap, join in ((True, True), (False, False), (True, False)):
        monkeypatch.setattr(SequentialEngine, "_overlap_tiles", overlap)
        monkeypatch.setattr(SequentialEngine, "_join_bwd", join)
        p = em.Parameters(periodicity=periodicity, batch_size=B, dist_sig_parameters=sig or (math.sqrt(2.0 * D), 12, 6, 1, 2, 6), n_steps=steps)
        e = em.EncoderMap(p, data, model=_weights(p, 2 * D if periodicity < INF else D), read_only=True)
        e._engine.run_steps(steps, torch.from_numpy(idx.reshape(-1)).cuda())
        loss = e._engine.losses()
        assert e._engine._tiles_ran == overlap and e._engine._bwd_ran == (overlap and join)
        out.append((loss, e._engine.w.cpu().numpy(), e._engine.loss_terms()["distance"], e._engine.sk_sync.cpu().numpy()))
    for k in (1, 2):
        np.testing.assert_array_equal(out[0][0], out[k][0])
        np.testing.assert_array_equal(out[0][1], out[k][1])
        np.testing.assert_array_equal(out[0][2], out[k][2])
    assert (out[0][3] == 0).all() and (out[2][3] == 0).all()      # the launches left their counters at zero, nobody gave up waiting
    assert np.isfinite(out[0][0]).all() and out[0][0][-1] < out[0][0][0]
    # the default choice of kernel for the separate launches: where it is the transposed chain (csrc/tchain.h), the same run to
    # float32 rounding -- losses to 2e-5 relative over the whole run, weights to 2e-4 of their scale (Adam's first steps are sign-like:
    # a gradient that differs in its last bits moves a weight by the full learning rate)
    from encodermap_amd import _lib as L
    monkeypatch.delenv("EMX_TCHAIN")
    monkeypatch.setattr(SequentialEngine, "_overlap_tiles", False)
    monkeypatch.setattr(SequentialEngine, "_join_bwd", False)
    c0 = L.load().emx_transposed_chain_count()
    p = em.Parameters(periodicity=periodicity, batch_size=B, dist_sig_parameters=sig or (math.sqrt(2.0 * D), 12, 6, 1, 2, 6), n_steps=steps)
    e = em.EncoderMap(p, data, model=_weights(p, 2 * D if periodicity < INF else D), read_only=True)
    e._engine.run_steps(steps, torch.from_numpy(idx.reshape(-1)).cuda())
    transposed = L.load().emx_transposed_chain_count() > c0
    assert transposed == (periodicity == INF and D <= 128 and B <= 16 * torch.cuda.get_device_properties(0).multi_processor_count)
    if transposed:
        assert not np.array_equal(e._engine.losses(), out[1][0])      # (it did run: another order of summation)
    np.testing.assert_allclose(e._engine.losses(), out[1][0], rtol=2e-5)
    w = e._engine.w.cpu().numpy()
    np.testing.assert_allclose(w, out[1][1], rtol=0, atol=2e-4 * np.abs(out[1][1]).max())


LOCAL_CASES = [
    # B, D, steps, periodicity, n_neurons
    (4096, 128, 12, INF, None), (1000, 128, 12, INF, None), (640, 32, 12, INF, None), (256, 10, 12, 2 * math.pi, None), (100, 33, 12, 2 * math.pi, None),
    (512, 64, 12, INF, [300, 40, 2]), (48, 8, 12, INF, [20, 3]), (16, 4, 6, INF, None), (2048, 100, 8, INF, [128, 128, 128, 2]),
    (512, 256, 8, INF, [256, 256, 2]),      # 272 blocks: more workgroups than CUs
]


@pytest.mark.parametrize("B,D,steps,periodicity,neurons", LOCAL_CASES)
def test_weight_gradients_without_slice_partials(em, B, D, steps, periodicity, neurons, monkeypatch):
    """emx_mlp_wgrad_adam_local (a workgroup per 32 x 32 block of a kernel over all rows, the optimizer in the same workgroup, the
    epilogue and the next batch's rows in the same launch) against emx_mlp_wgrad + emx_adam_step_slices: the gradient is summed in
    another order, so the runs agree to float32 rounding -- weights after ONE step to 2e-6 (relative to the update), the loss
    trajectory and every term to 2e-4 -- while everything that does not pass through the sums is identical: the rows gathered for
    the next step, the counters; and the launch is deterministic (two runs bit for bit)."""
    from encodermap_amd.engine import SequentialEngine

    rng = np.random.default_rng(B + D)
    data = rng.uniform(-math.pi, math.pi, (6000, D)).astype(np.float32) if periodicity < INF else blobs(rng, 6000, D)
    idx = rng.integers(0, 6000, (steps, B))
    kw = dict(n_neurons=neurons, activation_functions=[""] + ["tanh"] * (len(neurons) - 1) + [""]) if neurons else {}

    def run(local, n):
        monkeypatch.setattr(SequentialEngine, "_local_wgrad", local)
        p = em.Parameters(periodicity=periodicity, batch_size=B, dist_sig_parameters=(math.sqrt(2.0 * D), 12, 6, 1, 2, 6), n_steps=steps, **kw)
        e = em.EncoderMap(p, data, read_only=True, model=_weights(p, 2 * D if periodicity < INF else D))
        eng = e._engine
        w0 = eng.w.cpu().numpy().copy()
        eng.run_steps(n, torch.from_numpy(idx[:n].reshape(-1)).cuda())
        loss = eng.losses()
        assert eng._tail_ran == local and (eng.wg_sync.cpu().numpy() == 0).all()
        return dict(loss=loss, w=eng.w.cpu().numpy(), w0=w0, m=eng.m.cpu().numpy(), v=eng.v.cpu().numpy(), xb=eng.xb.cpu().numpy(), xin=eng.xin.cpu().numpy(),
                    counters=np.array([int(eng.step), int(eng.cursor), int(eng.hstep)]), terms=eng.loss_terms(), store=eng.plan.mirror.cpu().numpy())

    a1, b1 = run(True, 1), run(False, 1)
    upd = np.abs(b1["w"] - b1["w0"]).max()
    assert upd > 0
    assert np.abs(a1["w"] - b1["w"]).max() <= 2e-3 * upd          # (Adam's first step is lr * g / (|g| + eps): a rounding of g moves it by ~1e-7 / |g|)
    np.testing.assert_allclose(a1["m"], b1["m"], rtol=2e-5, atol=1e-9 + 5e-6 * np.abs(b1["m"]).max())     # (m = 0.1 clip(g): a sum of B products, rounded in another order)
    np.testing.assert_array_equal(a1["counters"], b1["counters"])
    np.testing.assert_array_equal(a1["loss"], b1["loss"])        # (the first step's cost does not see the optimizer)
    a, b, a2 = run(True, steps), run(False, steps), run(True, steps)
    for k in ("loss", "w", "m", "v", "xb", "xin", "store"):
        np.testing.assert_array_equal(a[k], a2[k])                 # deterministic
    np.testing.assert_array_equal(a["counters"], b["counters"])
    np.testing.assert_array_equal(a["xb"], b["xb"])
    np.testing.assert_array_equal(a["xin"], b["xin"])
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=2e-4)
    for k in a["terms"]:
        np.testing.assert_allclose(a["terms"][k], b["terms"][k], rtol=2e-4, atol=1e-6)
    assert np.isfinite(a["loss"]).all() and a["loss"][-1] < a["loss"][0]


def test_short_calls_equal_one_long_call(em):
    """run_steps appends the batches of its own shuffle stream behind the cursor, one batch ahead (no cursor reset, no head gather
    per call): any split of a run into calls -- single steps included, across the restart of the index block after ~1000 steps,
    with a caller's index block in between (the batch drawn ahead goes back to the stream) -- trains on the same batches in the
    same order as one call, bit for bit, and as the host statement of the stream (layout.BatchIndexStream) says."""
    from encodermap_amd.layout import BatchIndexStream

    rng = np.random.default_rng(12)
    data = rng.uniform(-math.pi, math.pi, (700, 6)).astype(np.float32)
    kw = dict(batch_size=64, seed=5)

    def engine(n):
        p = em.Parameters(n_steps=n, **kw)
        return em.EncoderMap(p, data, model=_weights(p, 12), read_only=True)._engine

    # (a) one call of 1130 steps (beyond the 1024 batches of the index block) against portions of 1, 7, 100, ... steps
    a = engine(1130)
    a.run_steps(1130)
    b = engine(1130)
    for k in (1, 7, 100, 1, 1, 400, 500, 20, 99, 1):
        b.run_steps(k)
    np.testing.assert_array_equal(a.losses(), b.losses())
    np.testing.assert_array_equal(a.w.cpu().numpy(), b.w.cpu().numpy())
    # (a') the same stream without graphs (eager launches) and in portions
    p = em.Parameters(n_steps=60, **kw)
    g = em.EncoderMap(p, data, model=_weights(p, 12), read_only=True, use_graph=False)._engine
    for k in (3, 1, 40, 16):
        g.run_steps(k)
    np.testing.assert_array_equal(a.losses()[:60], g.losses())
    # (b) own stream, a caller's block, own stream again == the same batches injected throughout
    twin = BatchIndexStream(700, 64, seed=b._shuffle_seed)
    own = twin.next_block(10)
    inj = rng.integers(0, 700, (3, 64))
    c = engine(13)
    c.run_steps(5)
    c.run_steps(3, torch.from_numpy(inj.reshape(-1)).cuda())
    c.run_steps(5)
    d = engine(13)
    d.run_steps(13, torch.from_numpy(np.concatenate([own[:5], inj, own[5:]]).reshape(-1)).cuda())
    np.testing.assert_array_equal(c.losses(), d.losses())
    np.testing.assert_array_equal(c.w.cpu().numpy(), d.w.cpu().numpy())
    # (c) a NEW capture two slots in front of the end of the index block (the history outgrown there): the capture's warm replays
    # run 32 + 2 batches ahead of the cursor -- into the zeroed slack slots behind the block's capacity, never past its end -- and
    # leave no trace: the run equals the one call of (a) bit for bit
    e = engine(1130)
    cap = e._block_capacity()
    e.run_steps(cap - 3)
    assert e._ring_pos == cap - 3 and e.idx_block.numel() == (cap + e._block_slack()) * e.B
    e._grow_history(e.history_capacity + 1)          # (drops the graphs: captured pointers are stale)
    assert e._graph is None
    e.run_steps(1130 - (cap - 3))
    np.testing.assert_array_equal(a.losses(), e.losses())
    np.testing.assert_array_equal(a.w.cpu().numpy(), e.w.cpu().numpy())
    assert int(e.idx_block[cap * e.B:].abs().max()) == 0


def _golden_module():
    import importlib.util

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_full_size_allpairs.py")
    spec = importlib.util.spec_from_file_location("make_full_size_allpairs", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name", ["C4", "C5"])
def test_full_size_all_pairs_pass_against_committed_float64_numbers(em, name):
    """The all-pairs (sketch-map) pass at the FULL batch of BASELINE configurations 4 (8192 x 512, periodic) and 5 (16384 x 256):
    cost, |dZ|, and 384 rows of d cost / d z against float64 numbers computed once in the build container by the direct-difference
    oracle (tests/golden/make_full_size_allpairs.py -> full_size_allpairs.npz; the seeded batch is re-generated here bit for
    bit), for every form of the pass that runs at this size: C5 -- 128 x 128 macro-tiles on split rows (the single process) and
    the dealt 64 x 64 Gram tiles of two row-block ranks; C4 -- VALU tiles on 32-bit phases, the float form, and dealt tiles.
    Tolerances: cost 1e-5 relative, gradient 2e-5 of its largest entry (the kernels' bounds at every other size)."""
    from encodermap_amd import _lib as L
    from encodermap_amd.engine import DataParallel

    G = _golden_module()
    B, Dd, Ld, per, sig, scale, _ = G.CONFIGS[name]
    x, z, rows = G.make_batch(name)
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_size_allpairs.npz"))
    assert np.array_equal(gold[f"{name}_rows"], rows)
    cost_o, norm_o, g_o, gmax = float(gold[f"{name}_cost"]), float(gold[f"{name}_dz_norm"]), gold[f"{name}_dz_rows"], float(gold[f"{name}_dz_absmax"])
    X, Z = torch.from_numpy(x).cuda(), torch.from_numpy(z).cuda()
    split = None
    if per == INF:
        split = torch.zeros(L.sketchmap_split_bytes(B, Dd), dtype=torch.uint8, device="cuda")
        L.sketchmap_split(X, split)
    n_ws = L.sketchmap_ws_bytes(B, Ld) // 4

    def check(label, cost, g):
        assert cost == pytest.approx(cost_o, rel=1e-5), label
        assert np.isfinite(g).all(), label
        np.testing.assert_allclose(g[rows], g_o, rtol=0, atol=2e-5 * gmax, err_msg=label)
        assert float(np.sqrt((g.astype(np.float64) ** 2).sum())) == pytest.approx(norm_o, rel=2e-5), label
        assert np.abs(g.astype(np.float64).sum(0)).max() < 2e-5 * gmax * math.sqrt(B), label     # zero net gradient

    def single(span):
        ws = torch.zeros(n_ws, dtype=torch.float32, device="cuda")
        dist, dZ = torch.zeros(1, device="cuda"), torch.full((B, Ld), float("nan"), device="cuda")
        L.sketchmap_tiles(X, Z, per, sig, 0, B, ws, split, data_span=span)
        L.sketchmap_finish(B, Dd, Z, sig, scale, 0.0, 0, B, dist, None, dZ, ws, periodicity=per)
        return float(dist), dZ.cpu().numpy()

    if per == INF:
        check("macro-tiles", *single(None))
    else:
        span = float(x.max() - x.min())
        assert span < 1.5 * per
        check("phase tiles", *single(span))
        check("float tiles", *single(None))
    # two row-block ranks: the dealt 64 x 64 tiles, shares summed
    loss_sum, g_sum = 0.0, np.zeros((B, Ld))
    for rank in range(2):
        dp = DataParallel(rank, 2, "rowblock")
        rb, re = dp.row_block(B)
        ws = torch.zeros(n_ws, dtype=torch.float32, device="cuda")
        dZ = torch.full((B, Ld), float("nan"), device="cuda")
        L.sketchmap_tiles_dealt(X, Z, per, sig, ws, split, rank, 2, data_span=(float(x.max() - x.min()) if per != INF else None))
        L.sketchmap_reduce_dealt(B, Dd, Z, per, sig, scale, 0.0, rb, re, rank, 2, dZ, ws)
        sk = L.sketchmap_describe(X, Z, per, sig, scale, 0.0, 0, B, ws)
        torch.cuda.synchronize()
        off = (sk.tile_loss - ws.data_ptr()) // 4
        loss_sum += float(ws[off:off + sk.n_tiles].double().sum()) * sk.loss_scale
        g_sum += dZ.cpu().numpy().astype(np.float64)
    check("dealt tiles, two ranks", loss_sum, g_sum)
