"""Data-parallel engine path on the GPU: two ranks (processes) share the one GPU of the test box and reduce through
gloo (RCCL refuses two ranks on one device; the collective is a single ``all_reduce`` of the flat buffer either way).
Exercises exactly what ``bench.py --gpus N`` runs: the step captured as two hipGraphs around the eager all-reduce.

* rowblock mode reproduces the single-process loss trajectory (exact partition of the objective)
* replica mode: both ranks hold identical weights after every step and the reported loss is the mean of their batches
"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PI = np.pi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(euclid=False):
    rng = np.random.default_rng(0)
    if euclid:      # Euclidean blobs: the all-pairs pass runs its 64 x 64 tiles on the f16 matrix cores
        return (rng.normal(0, 4, (8, 24))[rng.integers(0, 8, 3000)] + rng.normal(0, 1, (3000, 24))).astype(np.float32)
    return rng.uniform(-PI, PI, (3000, 12)).astype(np.float32)


def _params(em, euclid=False):
    if euclid:
        return em.Parameters(n_steps=6, batch_size=600, seed=11, periodicity=float("inf"), dist_sig_parameters=(8.0, 12, 6, 1, 2, 6))
    return em.Parameters(n_steps=6, batch_size=256, seed=11)


def _weights(p):
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights

    lay = build_layout(p.n_neurons, p.activation_functions, 24)
    return unpack_weights(lay, init_parameters(lay, 2))


def _worker(rank, world, port, mode, out_dir, euclid=False, collectives=None, p2p=False):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import encodermap_amd as em
    from encodermap_amd.engine import DataParallel

    data = _data(euclid)
    p = _params(em, euclid)
    e = em.EncoderMap(p, data, model=_weights(p), read_only=True, dp=DataParallel(rank, world, mode, collectives=collectives, p2p=p2p))
    assert e._engine._dealt() == (mode == "rowblock" and collectives == 2)
    if mode == "rowblock":
        idx = np.random.default_rng(5).integers(0, 3000, (6, p.batch_size))
        e._engine.run_steps(6, torch.from_numpy(idx.reshape(-1)).cuda())
    else:
        e._engine.run_steps(6)
    if p2p:
        assert len(e._engine._graph) == 1 and e._engine.dp._p2p_ctx and not e._engine.dp.p2p_gave_up()   # the collectives inside ONE step graph
    np.savez(os.path.join(out_dir, f"{mode}{'_p2p' if p2p else ''}_{rank}.npz"), loss=e._engine.losses(), w=e._engine.w.cpu().numpy(),
             terms=np.stack([e._engine.loss_terms()[k] for k in ("auto", "reg", "center", "distance")]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,world,collectives", [("replica", 2, None), ("replica", 3, None), ("rowblock", 2, 2), ("rowblock", 3, 1)])
def test_ranks_reduce_through_peer_mapped_memory(mode, world, collectives, tmp_path):
    """DataParallel(p2p=True): the step's collectives as the library's one-shot all-reduce (emx_p2p_*: every rank reads its peers'
    buffers through IPC mappings and adds them in rank order), captured inside the step's graph -- against the same ranks reducing
    through gloo: identical weights on every rank, and the two transports agree (bit for bit with two ranks: one addition either
    way; to rounding of the sums' order with three)."""
    import torch.multiprocessing as mp

    for p2p in (False, True):
        mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), True, collectives, p2p), nprocs=world, join=True)
    ref = [np.load(tmp_path / f"{mode}_{r}.npz") for r in range(world)]
    got = [np.load(tmp_path / f"{mode}_p2p_{r}.npz") for r in range(world)]
    for r in got[1:]:
        np.testing.assert_array_equal(got[0]["w"], r["w"])
        np.testing.assert_array_equal(got[0]["loss"], r["loss"])
    assert np.isfinite(got[0]["loss"]).all()
    if world == 2:
        np.testing.assert_array_equal(got[0]["loss"], ref[0]["loss"])
        np.testing.assert_array_equal(got[0]["w"], ref[0]["w"])
    else:
        np.testing.assert_allclose(got[0]["loss"], ref[0]["loss"], rtol=2e-5)
        diff = np.abs(got[0]["w"] - ref[0]["w"])
        assert (diff > 2e-5).mean() < 0.02 and diff.max() < 2e-3


@pytest.mark.parametrize("mode,world,euclid,collectives", [("rowblock", 2, False, 2), ("rowblock", 3, False, 2), ("rowblock", 2, True, 2),
                                                           ("rowblock", 3, True, 2), ("rowblock", 2, False, 1), ("rowblock", 3, True, 1),
                                                           ("rowblock", 2, True, 1), ("replica", 2, False, None)])
def test_ranks_on_one_gpu(mode, world, euclid, collectives, tmp_path):
    """rowblock: the dense layers on balanced runs of 16-row blocks (ragged at 3 ranks) and the all-pairs cost in both forms --
    two collectives: the tiles of the symmetric pass dealt round-robin, SUM all-reduces of the latent gradient and of
    [gradient | terms]; one collective: the rank's rows x all rows (SURVEY.md section 8e as written), the gradient of its own rows
    complete without an exchange -- the single-process trajectory either way."""
    import torch.multiprocessing as mp

    import encodermap_amd as em

    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), euclid, collectives), nprocs=world, join=True)
    runs = [np.load(tmp_path / f"{mode}_{r}.npz") for r in range(world)]
    for r in runs[1:]:
        np.testing.assert_array_equal(runs[0]["w"], r["w"])         # replicated optimizer: identical weights on every rank
        np.testing.assert_array_equal(runs[0]["loss"], r["loss"])
    assert np.isfinite(runs[0]["loss"]).all()
    if mode == "rowblock":
        data = _data(euclid)
        p = _params(em, euclid)
        e = em.EncoderMap(p, data, model=_weights(p), read_only=True)
        idx = np.random.default_rng(5).integers(0, 3000, (6, p.batch_size))
        e._engine.run_steps(6, torch.from_numpy(idx.reshape(-1)).cuda())
        np.testing.assert_allclose(runs[0]["loss"], e._engine.losses(), rtol=2e-5)
        single = np.stack([e._engine.loss_terms()[k] for k in ("auto", "reg", "center", "distance")])
        np.testing.assert_allclose(runs[0]["terms"], single, rtol=5e-5, atol=1e-9)
        diff = np.abs(runs[0]["w"] - e._engine.w.cpu().numpy())
        assert (diff > 2e-5).mean() < 0.02 and diff.max() < 2e-3


def _one_rank_rccl_worker(rank, port, out_dir, mode="replica", collectives=None):
    """The data-parallel path over RCCL itself, with the one rank a single GPU allows: two hipGraphs around the eager
    all-reduce, and (EMX_DP_GRAPH_ALLREDUCE=1) the all-reduce captured inside the step's graph."""
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    import encodermap_amd as em
    from encodermap_amd.engine import DataParallel, SequentialEngine

    class OneRank(DataParallel):
        @property
        def active(self):
            return True

    data = _data()
    out = {}
    for name, captured in (("eager", False), ("captured", True)):
        SequentialEngine._dp_graph_allreduce = captured
        p = em.Parameters(n_steps=20, batch_size=256, seed=11)
        e = em.EncoderMap(p, data, model=_weights(p), read_only=True, dp=OneRank(0, 1, mode, collectives=collectives))
        e._engine.run_steps(20)
        out[name] = e._engine.losses()
        out[name + "_graphs"] = np.array(len(e._engine._graph))
        out[name + "_w"] = e._engine.w.cpu().numpy()
    p = em.Parameters(n_steps=20, batch_size=256, seed=11)
    e = em.EncoderMap(p, data, model=_weights(p), read_only=True)
    e._engine.run_steps(20)
    out["single"] = e._engine.losses()
    np.savez(os.path.join(out_dir, "one_rank.npz"), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,collectives", [("replica", None), ("rowblock", 2), ("rowblock", 1)])
def test_one_rank_over_rccl(mode, collectives, tmp_path):
    """replica: one all-reduce per step; rowblock: the dealt all-pairs tiles (all of them: one rank), the latent-gradient
    all-reduce in front of the encoder backward and the [gradient | terms] all-reduce -- eagerly between three graphs, or both
    captured inside the step's graph."""
    import torch.multiprocessing as mp

    mp.spawn(_one_rank_rccl_worker, args=(_free_port(), str(tmp_path), mode, collectives), nprocs=1, join=True)
    r = np.load(tmp_path / "one_rank.npz")
    assert np.isfinite(r["eager"]).all() and len(r["eager"]) == 20
    assert int(r["eager_graphs"]) == (3 if collectives == 2 else 2) and int(r["captured_graphs"]) == 1
    np.testing.assert_array_equal(r["eager"], r["captured"])          # same kernels, same order: bitwise the same run
    np.testing.assert_array_equal(r["eager_w"], r["captured_w"])
    # one rank averages with itself: the single-process run up to the summation order of the gradient partials (the
    # data-parallel step reduces them in its own kernel), whose last-bit differences the trajectory amplifies
    if mode == "replica":   # (row-block ranks share one shuffle stream of their own: equality with the single-process run is
        #                      checked with injected batches in test_ranks_on_one_gpu)
        np.testing.assert_allclose(r["eager"][:3], r["single"][:3], rtol=1e-6)
        np.testing.assert_allclose(r["eager"], r["single"], rtol=1e-3)


@pytest.mark.parametrize("mode,extra,batch,scaling", [("replica", [], 2 * 4096, "weak"), ("rowblock", [], 4096, "strong"),
                                                       ("rowblock", ["--global-batch", "0"], 2 * 4096, "weak"),
                                                       # BASELINE config 5 (defined on 8 GPUs): ONE global batch of 16384 rows over the ranks
                                                       ("rowblock", ["--config", "C5", "--rows", "40000"], 16384, "strong")])
def test_bench_script_with_two_ranks(mode, extra, batch, scaling, tmp_path):
    """bench.py's N > 1 plumbing (rank environment, DataParallel modes, barrier, MAX over ranks of the timed region, one JSON
    line from rank 0 that names its objective and the ranks it saw) under torch.distributed.run with two ranks -- on this one
    GPU through its self-test hooks (EMX_BENCH_ONE_DEVICE, EMX_BENCH_BACKEND=gloo; the RCCL path itself needs two GPUs)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMX_BENCH_ONE_DEVICE="1", EMX_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "8",
           "--dp-mode", mode] + extra
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 24 and line["scaling"] == scaling
    cfg = line["config"]
    assert cfg["global_batch"] == batch and cfg["parallelism"] == f"dp2-{mode}" and cfg["ranks_seen"] == 2
    # (the form of the row-block step is chosen per configuration: C2-sized all-pairs passes as the rectangle with ONE collective,
    #  C5's as dealt tiles with two; replica ranks always one)
    #  (one 8192-row batch over TWO ranks: the rectangle would cost each rank half a 68-us pass more -- dealt tiles)
    assert cfg["collectives_per_step"] == (2 if ("C5" in extra or (mode == "rowblock" and batch == 8192)) else 1)
    assert cfg["allreduce"] == "eager between the step's graphs" and str(batch if mode == "rowblock" else 4096) in cfg["objective"]
    assert cfg["name"] == ("C5" if "C5" in extra else "C2") and cfg["name"] in cfg["workload"]
    assert line["value"] == pytest.approx(batch * line["steps_per_sec"], rel=1e-6)
    assert line["dtype"].startswith("f32") and line["roofline_all_pairs"]["kernel"] == "sk_tile_f16_kernel"
    assert all(np.isfinite(v) for v in line["loss_first_last"]) and "cpu_baseline" not in line


def test_bench_script_with_two_ranks_over_peer_mapped_memory(tmp_path):
    """bench.py --gpus 2 with EMX_DP_P2P=1 (the one-shot all-reduce without asking the probe; two ranks on this one GPU, gloo for the
    rendezvous and the timing's MAX): the line names the transport, the run trains."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EMX_BENCH_ONE_DEVICE="1", EMX_BENCH_BACKEND="gloo", EMX_DP_P2P="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "8"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["ranks_seen"] == 2 and cfg["allreduce"].startswith("one-shot over peer-mapped memory")
    assert cfg["p2p_allreduce_probe"] is None                      # (forced: nobody was asked)
    assert all(np.isfinite(v) for v in line["loss_first_last"]) and line["loss_first_last"][1] < line["loss_first_last"][0]


def test_bench_without_a_launcher_and_the_probe_decides(tmp_path):
    """(1) ``python bench.py --gpus 2`` with NO launcher and no rank environment -- the form the driver uses for one GPU -- starts its
    own ranks (two on this one GPU through the self-test hooks) and relays one line.  (2) The data-parallel path over RCCL with the
    one rank this box allows: the child-process probe (synthetic graph + the engine's captured data-parallel step) decides by
    itself that the all-reduce is captured in the step's graph -- no environment variable asks for it."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "EMX_DP_GRAPH_ALLREDUCE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "24", "--warmup", "8"], cwd=str(tmp_path),
                       env=dict(clean, EMX_BENCH_ONE_DEVICE="1", EMX_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["config"]["global_batch"] == 2 * 4096
    assert line["roofline"]["bound"] in ("issue", "valu", "hbm", "mfma") and line["roofline"]["limiter"]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--one-rank-dp", "--steps", "64", "--warmup", "32", "--no-cpu-baseline"],
                       cwd=str(tmp_path), env=dict(clean, MASTER_PORT=str(_free_port())), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["config"]["graph_allreduce_probe"] is True and line["config"]["allreduce"] == "captured in the step graph"
    assert line["config"]["backend"] == "nccl" and all(np.isfinite(v) for v in line["loss_first_last"])
