"""CPU tests of the host side: configuration objects, network layout, data generators, batch index stream,
checkpoint naming/resume rules, callback cadence, data-parallel partition, and the C-ABI library's exports."""
import ctypes
import json
import math
import os
import re

import numpy as np
import pytest

import encodermap_amd as em
from encodermap_amd import checkpoint as ckpt
from encodermap_amd.callbacks import CheckpointSaver, EncoderMapBaseCallback, ProgressBar
from encodermap_amd.layout import (BatchIndexStream, build_layout, init_parameters, l2_coefficients, pack_weights,
                                   unpack_weights, weights_from_initializer_dict)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INF = float("inf")


# ---- Parameters / ADCParameters (reference: encodermap/parameters/parameters.py) --------------------------
def test_parameter_defaults_match_reference_table():
    p = em.Parameters()
    # SURVEY.md appendix A (parameters.py:611-639)
    expect = dict(n_neurons=[128, 128, 2], activation_functions=["", "tanh", "tanh", ""], periodicity=2 * math.pi,
                  learning_rate=0.001, n_steps=1000, batch_size=256, summary_step=10, checkpoint_step=5000,
                  dist_sig_parameters=(4.5, 12, 6, 1, 2, 6), distance_cost_scale=500, auto_cost_scale=1,
                  auto_cost_variant="mean_abs", center_cost_scale=0.0001, l2_reg_constant=0.001, gpu_memory_fraction=0,
                  analysis_path="", id="", model_api="sequential", loss="emap_cost", training="auto", batched=True,
                  tensorboard=False, seed=None, current_training_step=0, write_summary=False,
                  trainable_dense_to_sparse=False, using_hypercube=False)
    assert p.defaults == expect
    for k, v in expect.items():
        assert p[k] == v and getattr(p, k) == v
    assert p.main_path == os.getcwd()
    a = em.ADCParameters()
    assert a.model_api == "functional" and a.auto_cost_scale is None and a.distance_cost_scale is None
    assert a.dihedral_cost_scale == 1 and a.angle_cost_scale == 0 and a.side_dihedral_cost_scale == 0.5
    assert a.cartesian_cost_scale_soft_start == (None, None) and a.cartesian_dist_sig_parameters == (4.5, 12, 6, 1, 2, 6)
    assert set(p.defaults) < set(a.defaults)
    # the WHOLE tables, keys and values, against the fixture AST-extracted from the reference's parameters.py:611-639, 794-828
    # (tests/golden/make_reference_kat_tables.py: data, no reference code)
    import json

    tables = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kat_tables.json")))["parameter_defaults"]

    def plain(v):
        if isinstance(v, dict) and "__tuple__" in v:
            return tuple(plain(x) for x in v["__tuple__"])
        if isinstance(v, dict) and "__float__" in v:
            return float(v["__float__"])
        return v

    for obj, name in ((p, "Parameters"), (a, "ADCParameters")):
        want = {k: plain(v) for k, v in tables[name].items()}
        assert set(obj.defaults) == set(want), (name, set(obj.defaults) ^ set(want))
        for k, v in want.items():
            assert obj.defaults[k] == v and type(obj.defaults[k]) is type(v), (name, k, obj.defaults[k], v)
            assert getattr(obj, k) == v, (name, k)


def test_parameter_behaviour(tmp_path, capsys):
    p = em.Parameters(batch_size=64, bogus=3, main_path=str(tmp_path))
    assert "Dropping unknown dict entry" in capsys.readouterr().out and not hasattr(p, "bogus")
    with pytest.raises(Exception, match="n_epochs"):
        em.Parameters(n_epochs=5)
    with pytest.raises(Exception, match="does not match"):
        em.Parameters(n_neurons=[10, 2], activation_functions=["", "tanh", "tanh", ""])
    with pytest.raises(Exception, match="tuple of len 2"):
        em.ADCParameters(cartesian_cost_scale_soft_start=(1, 2, 3))
    p["center_cost_scale"] = 2.5
    assert p.center_cost_scale == 2.5
    p.update(learning_rate=0.1, nonsense=1)
    assert p.learning_rate == 0.1
    path = p.save()
    assert path == str(tmp_path / "parameters.json")
    d = json.load(open(path))
    assert d["batch_size"] == 64 and d["main_path"] == str(tmp_path) and list(d) == sorted(d)   # sort_keys
    p.save()                                                     # second save backs the first one up
    assert len(list(tmp_path.glob("parameters_back_*.json"))) == 1
    q = em.Parameters.from_file(path)
    assert q.batch_size == 64 and q.center_cost_scale == 2.5 and q.dist_sig_parameters == [4.5, 12, 6, 1, 2, 6]
    moved = tmp_path / "sub"
    moved.mkdir()
    os.rename(path, moved / "parameters.json")
    r = em.Parameters.from_file(moved / "parameters.json")
    assert r.main_path == str(moved)
    y = p.save(tmp_path / "p.yaml")
    assert em.Parameters.load(y).batch_size == 64
    with pytest.raises(OSError):
        p.save(tmp_path / "p.txt")
    legacy = dict(p.to_dict(), n_epochs=3, n_steps_per_epoch=100)
    del legacy["n_steps"]
    json.dump(legacy, open(tmp_path / "legacy.json", "w"))
    lp = em.Parameters.from_file(tmp_path / "legacy.json")
    assert lp.n_steps == 300 and lp.summary_step == 1000
    assert "Non-standard value of batch_size" in str(p) and "Parameter" in em.Parameters.defaults_description()


# ---- layout (reference: models/models.py:3121-3220) -------------------------------------------------------
def test_layout_default_and_custom():
    lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], 20)
    assert [l.name for l in lay.layers] == ["Encoder_0", "Encoder_1", "Latent", "Decoder_0", "Decoder_1", "Decoder_2"]
    assert [(l.n_in, l.n_out) for l in lay.layers] == [(20, 128), (128, 128), (128, 2), (2, 128), (128, 128), (128, 20)]
    assert [l.activation for l in lay.layers] == ["tanh", "tanh", None, "tanh", "tanh", None]
    assert lay.n_enc == 3 and lay.latent_dim == 2 and lay.n_real == 38934            # P = 38 934
    assert build_layout([128, 128, 2], ["", "tanh", "tanh", ""], 128).n_real == 66690
    assert all(l.w_off % 4 == 0 for l in lay.layers) and lay.n_params % 4 == 0
    small = build_layout([64, 3], ["", "tanh", ""], 7)
    assert [(l.n_in, l.n_out, l.activation) for l in small.layers] == [(7, 64, "tanh"), (64, 3, None), (3, 64, "tanh"), (64, 7, None)]
    # any Keras activation whose derivative is a function of the layer's output (models.py:3179-3220 hands the name to Dense)
    assert [l.activation for l in build_layout([8, 4, 2], ["", "relu", "softplus", ""], 4).layers] == ["relu", "softplus", None, "softplus", "relu", None]
    assert [l.activation for l in build_layout([8, 2], ["", "linear", ""], 4).layers] == [None] * 4
    with pytest.raises(NotImplementedError, match="pre-activation"):
        build_layout([8, 2], ["", "gelu", ""], 4)


def test_initialisers_and_packing():
    lay = build_layout([128, 128, 2], ["", "tanh", "tanh", ""], 64)
    flat = init_parameters(lay, 5)
    assert np.array_equal(flat, init_parameters(lay, 5)) and not np.array_equal(flat, init_parameters(lay, 6))
    arrays = unpack_weights(lay, flat)
    W1, b1 = arrays[2], arrays[3]
    std = math.sqrt(1 / 128) / 0.87962566103423978
    assert np.abs(W1).max() <= 2 * std + 1e-7                       # truncated at 2 sigma
    assert abs(W1.std() - math.sqrt(1 / 128)) < 0.003               # VarianceScaling: var = 1/fan_in after truncation
    assert abs(b1.mean() - 0.1) < 0.02 and abs(b1.std() - 0.05) < 0.02
    assert np.array_equal(pack_weights(lay, arrays), flat)
    c = l2_coefficients(lay, 0.001)
    assert c.sum() == pytest.approx(0.002 * sum(l.n_in * l.n_out for l in lay.layers), rel=1e-6)
    d = {}
    for i in range(6):
        stem = "dense" if i == 0 else f"dense_{i}"
        d[f"{stem}/kernel"], d[f"{stem}/bias"] = arrays[2 * i], arrays[2 * i + 1]
    assert np.array_equal(weights_from_initializer_dict(lay, d), flat)
    with pytest.raises(ValueError):
        pack_weights(lay, arrays[:-1])


def test_feistel_permutation_is_a_keyed_bijection():
    """Host statement of emx_permute (include/emx.h): a permutation for every n, a different one per seed and per epoch,
    reproducible, and without visible structure (rank correlation with the identity ~ n^-1/2, no fixed-point excess)."""
    from encodermap_amd.layout import feistel_permutation

    for n in (1, 2, 3, 5, 16, 17, 255, 256, 257, 1000, 65537, 100003):
        p = feistel_permutation(n, 11, 0)
        assert p.dtype == np.int64 and np.array_equal(np.sort(p), np.arange(n)), n
    a, b, c = feistel_permutation(50000, 7, 0), feistel_permutation(50000, 7, 1), feistel_permutation(50000, 8, 0)
    assert np.array_equal(a, feistel_permutation(50000, 7, 0))
    assert (a == b).mean() < 1e-3 and (a == c).mean() < 1e-3 and (a == np.arange(50000)).mean() < 1e-3
    assert abs(np.corrcoef(np.arange(50000), a)[0, 1]) < 0.02
    # batches drawn from it are balanced: every 1000-row window of the permutation covers the index range evenly
    assert abs(a[:1000].mean() / 25000 - 1) < 0.1


def test_batch_index_stream_semantics():
    s = BatchIndexStream(10, 4, seed=0)
    blk = s.next_block(5)                       # 20 indices = exactly two passes
    assert blk.shape == (5, 4)
    flat = blk.reshape(-1)
    assert sorted(flat[:10]) == list(range(10)) and sorted(flat[10:]) == list(range(10))   # one permutation per pass
    assert set(blk[2]) & set(flat[:10]) and set(blk[2]) & set(flat[10:])                  # batch 2 straddles the passes
    nxt = s.next_block(1)
    assert nxt.shape == (1, 4)


# ---- misc data generators -----------------------------------------------------------------------------------
def test_create_n_cube_matches_networkx_edge_order():
    nx = pytest.importorskip("networkx")
    from encodermap_amd.misc import _hypercube_edges

    for n in (1, 2, 3, 4, 5):
        ref = np.array([e for e in nx.hypercube_graph(n).edges]).reshape(-1, 2, n)
        assert np.array_equal(_hypercube_edges(n), ref)
    coords, colors = em.create_n_cube(2, sigma=0)
    assert np.array_equal(coords[0], [0.0, 0.0])                    # misc/misc.py doctest
    c4, _ = em.create_n_cube(4, seed=3)
    assert c4.shape == (16000, 4)                                    # autoencoder.py:654-655
    assert np.array_equal(c4, em.create_n_cube(4, seed=3)[0])
    pts, ids = em.random_on_cube_edges(5000, sigma=0.0, rng=np.random.default_rng(0))
    on_edge = ((np.abs(pts) < 1e-12) | (np.abs(pts - 1) < 1e-12)).sum(1) >= 2
    assert on_edge.all() and set(ids) == set(range(12))


# ---- checkpoints ---------------------------------------------------------------------------------------------
def test_checkpoint_roundtrip_and_rules(tmp_path):
    lay = build_layout([16, 2], ["", "tanh", ""], 6)
    arrays = unpack_weights(lay, init_parameters(lay, 1))
    state = {"m": np.ones(lay.n_params, np.float32), "v": np.full(lay.n_params, 2, np.float32), "step": 20}
    for s in (10, 20, 5):
        ckpt.save_checkpoint(ckpt.checkpoint_name(tmp_path, s), arrays, dict(state, step=s), "EncoderMap")
    assert [ckpt.step_of(f) for f in ckpt.find_checkpoints(tmp_path)] == [5, 10, 20]
    assert ckpt.resolve(tmp_path).name == "saved_model_20.emx.npz"
    payload = ckpt.load_checkpoint(ckpt.resolve(tmp_path))
    assert np.array_equal(payload["dense/kernel"], arrays[0]) and np.array_equal(payload["dense_3/bias"], arrays[7])
    assert int(payload["adam/iterations"]) == 20 and str(payload["class_name"]) == "EncoderMap"
    with pytest.raises(Exception, match="has no saved_model"):
        ckpt.resolve(tmp_path / "..".replace("..", "empty")) if (tmp_path / "empty").mkdir() is None else None
    p = em.Parameters(main_path=str(tmp_path), current_training_step=20, n_steps=20)
    p.save()
    assert ckpt.reconcile_step(tmp_path / "saved_model_20.emx.npz", p, False) is p
    with pytest.raises(Exception, match="intermediate"):
        ckpt.reconcile_step(tmp_path / "saved_model_10.emx.npz", p, False)
    p2 = ckpt.reconcile_step(tmp_path / "saved_model_10.emx.npz", p, True)
    assert p2.current_training_step == 10 and p2.n_steps == 10      # non-ADC branch: no +1 (saving_loading_models.py:614-615)
    pa = em.ADCParameters(main_path=str(tmp_path), current_training_step=20, n_steps=20)
    pa.save()
    pa2 = ckpt.reconcile_step(tmp_path / "saved_model_10.emx.npz", pa, True)
    assert pa2.current_training_step == 11 and pa2.n_steps == 11    # ADC branch: step + 1 (saving_loading_models.py:557-558)
    assert list(tmp_path.glob("parameters_at_20_*")) and list(tmp_path.glob("saved_model_20_backup_from_10_*"))
    stamped = ckpt.checkpoint_name(tmp_path, None)
    assert re.fullmatch(r"saved_model_\d{4}-\d\d-\d\dT.*\.emx\.npz", stamped.name)


# ---- callbacks ---------------------------------------------------------------------------------------------------
def test_callback_cadence():
    p = em.Parameters(summary_step=10, checkpoint_step=25)
    log = []

    class CB(EncoderMapBaseCallback):
        def on_summary_step(self, step, logs=None):
            log.append(("s", step))

        def on_checkpoint_step(self, step, logs=None):
            log.append(("c", step))

    cb = CB(p)
    for b in range(50):
        cb.on_train_batch_end(b)
    assert log == [("s", 10), ("s", 20), ("c", 25), ("s", 30), ("s", 40), ("c", 50), ("s", 50)]   # checkpoint before summary
    assert cb.due_steps(0, 50) == [10, 20, 25, 30, 40, 50] and cb.due_steps(25, 30) == [30]
    assert CheckpointSaver(p).due_steps(0, 60) == [25, 50]
    assert ProgressBar(p).due_steps(0, 100) == [] and ProgressBar(p, enabled=True).due_steps(0, 20) == [10, 20]


# ---- data-parallel partition -----------------------------------------------------------------
def test_row_block_partition():
    from encodermap_amd.engine import DataParallel

    for B, G in [(4096, 8), (4096, 3), (80, 3), (100, 7), (16, 1)]:
        blocks = [DataParallel(r, G, "rowblock").row_block(B) for r in range(G)]
        assert blocks[0][0] == 0 and blocks[-1][1] == B
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
        sizes = [e - b for b, e in blocks]
        assert max(sizes) - min(sizes) <= 16                      # balanced to one 16-row block ...
        assert all(b % 16 == 0 or b == B for b, _ in blocks)      # ... the granule of the weight-gradient fragments (emx_mlp_aop_bytes)
    # fewer 16-row blocks than ranks: the SAME clear error on every rank (a lone failing rank would leave the others in the all-reduce)
    for B, G in [(10, 4), (40, 4), (7, 7)]:
        for r in range(G):
            with pytest.raises(ValueError, match="at least one 16-row block per rank"):
                DataParallel(r, G, "rowblock").row_block(B)
    assert DataParallel(2, 8, "replica").row_block(4096) == (0, 4096)
    assert DataParallel().row_block(256) == (0, 256) and not DataParallel().active
    assert DataParallel(0, 2, "rowblock").batch_seed(5) == DataParallel(1, 2, "rowblock").batch_seed(5)
    assert DataParallel(0, 2, "replica").batch_seed(5) != DataParallel(1, 2, "replica").batch_seed(5)
    with pytest.raises(ValueError):
        DataParallel(0, 2, "pipeline")


# ---- the product path fails loudly without a GPU / library -----------------------------------------------------------
def test_windows_of_a_deep_network():
    """more than twelve Dense layers: windows of up to twelve, broken at the encoder / decoder boundary (encodermap_amd/_lib.py)"""
    from encodermap_amd._lib import EMX_MAX_LAYERS, _window_bounds

    assert EMX_MAX_LAYERS == 12
    assert _window_bounds(16, 8) == [(0, 8), (8, 16)]
    assert _window_bounds(28, 14) == [(0, 12), (12, 14), (14, 26), (26, 28)]
    assert _window_bounds(13, None) == [(0, 12), (12, 13)]
    assert _window_bounds(26, 1) == [(0, 1), (1, 13), (13, 25), (25, 26)]
    for n, n_enc in [(13, 6), (40, 20), (25, 24)]:
        w = _window_bounds(n, n_enc)
        assert w[0][0] == 0 and w[-1][1] == n and all(a[1] == b[0] for a, b in zip(w, w[1:]))
        assert all(0 < b - a <= 12 and not (a < n_enc < b) for a, b in w)


def test_no_silent_cpu_fallback():
    import torch

    from encodermap_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EmxError, match="no CPU fallback"):
        em.EncoderMap(em.Parameters(), np.zeros((300, 3), np.float32), read_only=True)
    # nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, "encodermap_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert not re.search(r"^\s*(from|import)\s+oracle", open(os.path.join(pkg, fn)).read(), re.M), fn


def test_library_exports_every_declared_symbol():
    from encodermap_amd import _lib

    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "emx.h")).read(), flags=re.S)
    declared = set(re.findall(r"^(?:int|int64_t|size_t|void|const char\*)\s+(emx_\w+)\s*\(", header, re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if not os.path.isfile(_lib.LIB_PATH):
        pytest.skip("libemx.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    loaded = _lib.load()
    assert loaded.emx_version() == 1 and loaded.emx_target_arch() == b"gfx950"
    # argument counts of the binding agree with the header
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", header, re.S)
        params = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        assert len(params) == len(args), (name, len(params), len(args))


def test_torsion_table_far_sides_and_bit_layout():
    """Host side of the all-atom back-mapping: the graph cut (misc/rotate.py:409-511) against the oracle's, the lane-major
    bit table of include/emx.h, and the reference's error behaviour for bonds that are missing or sit in a ring."""
    from molecules import random_molecule
    from encodermap_amd.backmapping import AllAtomTopology, TorsionTable
    from oracle import allatom_ops as O

    xyz, bonds, cq, sq = random_molecule(700, 5, side_every=2)     # > 2048 atoms: two mask words per lane
    A = len(xyz)
    assert A > 2048
    quads = np.concatenate([cq[:40], sq[:25]])
    t = TorsionTable(A, bonds, quads)
    m = t.packed_far_mask()
    assert m.shape == (len(quads), 2, 64) and m.dtype == np.uint32
    for j in (0, 7, 39, 40, 64):
        far = O.far_side(A, bonds, quads[j][1:3])
        assert np.array_equal(np.sort(t.far_sides[j]), far)
        atoms = np.arange(A)
        bit = (m[j][(atoms // 64) // 32, atoms % 64] >> ((atoms // 64) % 32).astype(np.uint32)) & 1
        assert np.array_equal(np.flatnonzero(bit), far)
    active = np.ones(len(quads), bool)
    active[3] = False
    assert TorsionTable(A, bonds, quads, active).n_torsions == len(quads) - 1
    with pytest.raises(Exception, match="not part of the graph"):
        TorsionTable(A, bonds, [[cq[0][0], cq[0][1], cq[0][3], cq[0][2]]])
    ring = np.concatenate([bonds, [[cq[0][0], cq[0][3]]]])
    with pytest.raises(Exception, match="still connected"):
        TorsionTable(A, ring, cq[:1])
    with pytest.raises(Exception, match="more than one connected component"):
        TorsionTable(A, bonds[1:], cq[:1])
    top = AllAtomTopology(xyz, bonds, cq, sq)
    assert (top.n_central, top.n_side) == (len(cq), len(sq)) and top.table.n_torsions == len(cq) + len(sq)


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` the way the driver calls it for one GPU -- no launcher, no environment: the script must start
    torch.distributed.run itself (a child process; nothing has touched a GPU), relay exactly ONE JSON line on stdout and the exit
    code, and repeat the run once with the eager all-reduce when the first attempt fails (--launch-check: the ranks meet over gloo
    and do no GPU work, so this runs on the build box)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "EMX_DP_GRAPH_ALLREDUCE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check", "--steps", "7", "--warmup", "3"],
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 7 and line["warmup"] == 3 and line["value"] == 3.0
    assert line["config"]["ranks_seen"] == 2
    # (the first attempt, as a rule; a rendezvous that loses its port to another test's lingering socket is exactly what the retry
    #  is for -- then the launcher must have said so)
    assert line["config"]["allreduce"] == "probe decides" or (line["config"]["allreduce"] == "eager (forced)"
                                                              and "repeating it with the eager all-reduce" in r.stderr), r.stderr[-2000:]
    assert line["config"]["probe_port_agreed"] is True      # (rank 0's free port, learnt by every rank over a short-lived gloo group)
    assert "a stray line from a rank" in r.stderr          # (relayed, but not on stdout)
    # the ranks under the launcher DIRECTLY (the driver's form for N > 1), one of them never coming back: the watchdog ends the run
    # with a reason and a non-zero code instead of a hang
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], cwd=str(tmp_path),
                       env=dict(env, EMX_BENCH_HANG="1", EMX_BENCH_WATCHDOG="3"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "did not finish within 3 s" in r.stderr and '"metric"' not in r.stdout
    # first attempt fails on every rank -> second attempt with the eager all-reduce forced
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], cwd=str(tmp_path),
                       env=dict(env, EMX_BENCH_FAIL_CAPTURED="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["allreduce"] == "eager (forced)" and "repeating it with the eager all-reduce" in r.stderr
    # both attempts fail: the exit code is relayed, stdout stays empty
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], cwd=str(tmp_path),
                       env=dict(env, EMX_BENCH_FAIL_CAPTURED="1", EMX_DP_GRAPH_ALLREDUCE="0", WORLD_SIZE="3", RANK="0"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_bench_quotes_counters_of_this_build_only(monkeypatch):
    """bench.py takes HBM traffic and in-step durations from the committed PMC passes only when they were taken on THESE kernel sources
    (sha256 over csrc/* and include/*.h): a counters file of another build is refused with a reason, for the step's launches and for
    the inference launch alike; the step's listing holds the step's launches only."""
    import bench

    h = bench.source_hash()
    assert len(h) == 16 and h == bench.source_hash()
    blob, path, reason = bench._counters_file("C2")
    if blob is not None:                                   # (the committed file is this build's: it is quoted)
        assert blob["source_hash"] == h and reason is None
        n, src = bench._inference_traffic("C2", "encode")
        assert n is None or (n > 5e8 and "infer16" in src)     # 1e6 x 128 rows in: at least the algorithmic 520 MB

        class E:
            B, periodic = 4096, False

        listing, _ = bench.roofline_step("C2", E(), {})
        assert listing and not any("infer16" in r["kernel"] for r in listing)
    else:
        assert reason.startswith("refused") or reason.startswith("no counters")
    monkeypatch.setattr(bench, "source_hash", lambda: "0123456789abcdef")
    blob, path, reason = bench._counters_file("C2")
    assert blob is None and reason.startswith("refused") and "0123456789abcdef" in reason
    assert bench._inference_traffic("C2", "decode")[0] is None and bench._inference_traffic("C4", "encode")[0] is None
    assert bench.roofline_step("C2", None, {})[0] is None
