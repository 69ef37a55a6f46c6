#!/usr/bin/env python3
"""Headline benchmark: EncoderMap training throughput at B=4096, D=128 (BASELINE.json config C2).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full training step of ``em.EncoderMap`` (gather a mini-batch, encoder/decoder forward, sketch-map
distance + auto + center + L2 cost, backward, clip + Adam) on synthetic Gaussian blobs that are resident in HBM
before the timed region starts.  One JSON line is printed by rank 0:

  value        whole-job samples/s = N_gpus * 4096 * K / (max over ranks of the time for K steps)
  roofline     the kernel that takes the largest share of the STEP (today the fused dense-layer launch: forward, auto cost,
               decoder backward), ``roofline_all_pairs`` the O(B^2 D) all-pairs kernel.  Both are timed here with HIP events
               around back-to-back launches on the live batch.  Every float32 product of these kernels runs as THREE f16
               matrix instructions on operands split into two 16-bit planes, so ``achieved`` counts the executed FLOPs
               (3 x the algorithmic 2 per multiply-add) and ``peak`` is the dense f16 matrix peak; ``fractions`` gives the
               three ways to price the same launch: fp32_equivalent (algorithmic FLOPs against the 157.3 TFLOP/s fp32 matrix
               peak -- the precision delivered), executed_f16 (= ``frac``) and valu_issue (VALU instructions per launch, from the
               PMC passes under profiles/, per SIMD and clock -- what actually bounds both kernels).
               ``traffic`` = HBM bytes per launch from the PMC passes under profiles/ (``traffic_source`` names the file).
  cpu_baseline the reference-formulation CPU path (oracle/reference_ops.py in float32: Gram-matrix distances with
               B x B temporaries, autograd, per-element clipped Adam) timed on this host for a few steps of the same
               workload.  N=1 only.  kind "port": TensorFlow itself cannot be installed here.

N > 1: weak scaling -- every rank trains on its own mini-batch of 4096 rows (global batch 4096*N) and the flat
gradient buffer is averaged by one RCCL all-reduce per step (engine.DataParallel mode "replica": the objective is the
mean of N independent 4096-row sketch-map costs, what N reference processes with averaged gradients would minimise).
``--dp-mode rowblock`` is the other objective: ONE sketch-map cost over ONE global batch (``--global-batch``, default
4096: strong scaling of the single-GPU objective; ``--global-batch 0`` = 4096 * N), dense layers on runs of row blocks,
the symmetric all-pairs tiles dealt to the ranks, two all-reduces per step (latent gradient shares, [gradient | terms]).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, D, N_ROWS = 4096, 128, 1_000_000
SIG = (16.0, 12, 6, 1, 2, 6)
FP32_PEAK = 157.3e12   # MI355X fp32 vector == fp32 MFMA peak (MI355X_MICROARCH.md chip table)
F16_PEAK = 2.5e15      # dense f16 / bf16 matrix peak (same table; the headline figure with 2:1 sparsity is not used)
SIMDS, CLOCK = 1024, 2.4e9
DTYPE = "f32 (split-f16 x3 mfma)"
COUNTERS = os.path.join("profiles", "r02_step_counters.json")


def make_blobs(device, n_rows, d, seed):
    """C2 data: 16 blob centres N(0,4), points = centre + N(0,1)   (SURVEY.md section 8d)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    centres = torch.randn(16, d, generator=g, device=device) * 4.0
    k = torch.randint(0, 16, (n_rows,), generator=g, device=device)
    return centres[k] + torch.randn(n_rows, d, generator=g, device=device)


def cpu_baseline(n_steps=12):
    import numpy as np
    import torch

    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights
    from encodermap_amd.parameters import Parameters
    from oracle import reference_ops as R

    rng = np.random.default_rng(1)
    n_rows = 20000   # the sample: rows actually touched by the timed steps are drawn from this subset
    centres = rng.normal(0, 4, (16, D))
    data = (centres[rng.integers(0, 16, n_rows)] + rng.normal(0, 1, (n_rows, D))).astype(np.float32)
    p = Parameters(periodicity=float("inf"), batch_size=B, dist_sig_parameters=SIG)
    lay = build_layout(p.n_neurons, p.activation_functions, D)
    w0 = unpack_weights(lay, init_parameters(lay, 0))
    model = R.SequentialOracle(D, p.n_neurons, p.activation_functions, p.periodicity, w0, dtype=torch.float32)
    idx = rng.integers(0, n_rows, (n_steps + 2, B))
    R.train_steps(model, data, idx[:1], p.to_dict())           # warm-up step
    # torch-CPU over-subscribes badly on many-core hosts (128 threads are ~2.5x slower than 32 here), so the thread count
    # is chosen by a one-step probe among {all, 64, 32, 16} and the fastest setting is used and reported
    cores = os.cpu_count() or torch.get_num_threads()
    best, best_t = torch.get_num_threads(), float("inf")
    for nt in sorted({cores, min(cores, 64), min(cores, 32), min(cores, 16)}, reverse=True):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        R.train_steps(model, data, idx[1:2], p.to_dict())
        dt1 = time.perf_counter() - t0
        if dt1 < best_t:
            best, best_t = nt, dt1
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    R.train_steps(model, data, idx[2:], p.to_dict())
    dt = time.perf_counter() - t0
    return {"value": B * n_steps / dt, "unit": "samples/s", "cores": best, "kind": "port",
            "sample": f"{n_steps} training steps at B={B}, D={D} (float32 torch-CPU restatement of the reference formulation, "
                      f"{dt / n_steps * 1e3:.0f} ms/step with {best} of {cores} host threads, the fastest of a one-step probe)"}


def _time_launches(fn, launches):
    """average duration of ``fn`` (asynchronous launches on torch's current stream) over back-to-back calls, seconds"""
    import torch

    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / launches


def _counters(kernel_key):
    """static per-launch counters of the step's kernels (rocprofv3 PMC passes over this script, tools/run_pmc_step.sh)"""
    try:
        for name, c in json.load(open(os.path.join(ROOT, COUNTERS)))["kernels"].items():
            if kernel_key in name:
                return c
    except Exception:
        pass
    return {}


def _record(kernel, kernel_key, t, mults, note, algorithmic_bytes):
    """One roofline object: ``mults`` multiply-adds of float32 precision per launch, each executed as three f16 matrix
    instructions' worth of FLOPs."""
    c = _counters(kernel_key)
    executed = 6.0 * mults
    valu = c.get("SQ_INSTS_VALU")
    out = {"bound": "mfma", "achieved": executed / t / 1e12, "peak": F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": executed / t / F16_PEAK,
           "traffic": c.get("hbm_bytes_per_launch"), "traffic_source": f"{COUNTERS} (static: PMC passes over this workload)" if c else None,
           "kernel": kernel, "kernel_us": t * 1e6, "algorithmic_bytes": algorithmic_bytes,
           "fractions": {"fp32_equivalent": 2.0 * mults / t / FP32_PEAK, "executed_f16": executed / t / F16_PEAK,
                         "valu_issue": (valu / (SIMDS * CLOCK * t)) if valu else None},
           "fp32_equiv_tflops": 2.0 * mults / t / 1e12,
           "limiter": "instruction issue: neither the matrix pipes nor HBM are near their roofs; valu_issue is VALU wave-instructions "
                      "per SIMD and clock (measured ceiling 0.32-0.37 with >= 4 waves per SIMD, profiles/r01_valu_issue_rates.txt)",
           "note": note}
    return out


def kernel_rooflines(eng, launches):
    """(dominant kernel of the step, all-pairs kernel), each timed alone on the engine's live batch."""
    from encodermap_amd import _lib

    lay = eng.layout
    n = len(lay.layers)
    Z = eng.h[lay.n_enc - 1]
    Bt = eng.B
    t_sk = _time_launches(lambda: _lib.sketchmap_tiles(eng.xb, Z, float("inf"), SIG, 0, Bt, eng.sk_ws, eng.split_ws), launches)
    W = D * Bt * (Bt - 1) / 2          # pair-elements per launch (i<->j symmetry exploited): one multiply-add each in the Gram form
    sk = _record("sk_tile_f16_kernel", "sk_tile_f16", t_sk, W,
                 "Gram tiles on rows split into two f16 planes, VALU sigmoid epilogue (~40 instructions per pair); 2 FLOP per "
                 "pair-element over the upper triangle (SURVEY.md section 8d)", 4 * Bt * (D + 2 * 2 + 2) + 4)
    recs = [sk]
    if (eng.rb, eng.re) == (0, Bt) and eng._forward():      # one fused launch (single process / replica ranks)
        t_ch = _time_launches(eng._forward, launches)
        dims = [(l.n_in, l.n_out) for l in lay.layers]
        mults = Bt * (sum(k * m for k, m in dims) + sum(k * m for k, m in dims[lay.n_enc:]))   # forward + the decoder's dX products
        n_par = sum(k * m + m for k, m in dims)
        recs.append(_record("chain_kernel<FUSED>", "chain_kernel<FUSED>", t_ch, mults,
                            "forward of all layers, auto cost and the decoder's backward pass in one launch per 16-row block; weights "
                            "from the pre-split operand store, activations split in the epilogue; algorithmic bytes: the batch in, "
                            "activations + deltas out as f16 fragments for the weight-gradient kernel, one pass over the weights",
                            4 * Bt * D + 4 * Bt * sum(m for _, m in dims) * 2 + 4 * n_par))
    recs.sort(key=lambda r: -r["kernel_us"])
    return recs[0], sk


def probe_graph_allreduce(world: int, timeout_s: float = 120.0) -> bool:
    """N > 1: one child process per rank checks, BEFORE this process touches the GPU, whether an RCCL all-reduce captured in
    a hipGraph works with this many ranks on this machine (encodermap_amd/_probe_graph_allreduce.py).  Passed on every
    rank: the data-parallel step runs as one graph, eight steps per replay; failed, or hung until the
    timeout (the child is killed, its queues go with it): the all-reduce stays eager between two graphs."""
    import subprocess

    # the children rendezvous among themselves: nothing of the launcher's agent store (TORCHELASTIC_USE_AGENT_STORE makes
    # rank 0 a CLIENT of the launcher's store, which does not listen on the probe's port)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["EMX_PROBE_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 173)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    try:
        proc = subprocess.Popen([sys.executable, "-m", "encodermap_amd._probe_graph_allreduce"], cwd=ROOT, env=env,
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:
        return False
    try:
        return proc.wait(timeout=timeout_s) == 0
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dp-mode", default="replica", choices=["replica", "rowblock"])
    ap.add_argument("--global-batch", type=int, default=B,
                    help="--dp-mode rowblock: rows of the ONE global batch all ranks share (0: 4096 per rank)")
    ap.add_argument("--one-rank-dp", action="store_true",
                    help="N = 1 only: run the DATA-PARALLEL code path (probe, RCCL group, all-reduce) with its one rank -- a self-test, "
                         "not a benchmark configuration")
    args = ap.parse_args()

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # self-test hooks (not benchmark configurations): EMX_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and EMX_BENCH_BACKEND=gloo
    # replaces RCCL, so that the N > 1 plumbing of this script can be run with two ranks on a one-GPU box
    one_device = os.environ.get("EMX_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("EMX_BENCH_BACKEND", "nccl")
    device = "cuda:0" if one_device else f"cuda:{local_rank}"
    multi = world > 1 or (args.one_rank_dp and world == 1)
    if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
        del os.environ["NCCL_DEBUG"]   # (RCCL's version banner goes to stdout, which is reserved for the JSON line)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("LOCAL_RANK", "0")
    # The all-reduces run EAGERLY between the step's graphs unless EMX_DP_GRAPH_ALLREDUCE=1 asks for the captured form (one graph
    # per step, eight steps per replay): that form has only ever run with a one-rank RCCL group, so it is opt-in until a run
    # on two or more GPUs has validated it; asked for, it is still taken only if the child-process probe passes on every rank.
    probe = None
    graph_allreduce = False
    if multi and backend == "nccl" and os.environ.get("EMX_DP_GRAPH_ALLREDUCE") == "1":
        probe = probe_graph_allreduce(world)
        graph_allreduce = probe
    torch.cuda.set_device(device)
    import torch.distributed as dist

    if multi:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import encodermap_amd as em
    from encodermap_amd.engine import DataParallel, SequentialEngine

    if multi:
        # every rank must take the same path: the captured all-reduce only if the probe passed on ALL ranks
        flag = torch.tensor([1.0 if graph_allreduce else 0.0], device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        graph_allreduce = bool(float(flag) == 1.0)
        SequentialEngine._dp_graph_allreduce = graph_allreduce

    rowblock = args.dp_mode == "rowblock" and world > 1
    batch = (args.global_batch or B * world) if rowblock else B     # rows per engine step: the rank's own batch, or the shared one
    p = em.Parameters(periodicity=float("inf"), batch_size=batch, dist_sig_parameters=SIG, n_steps=args.steps + args.warmup, seed=7)
    dp = DataParallel(rank, world, args.dp_mode) if world > 1 else None
    if multi and world == 1:
        class _OneRank(DataParallel):      # the data-parallel path with the one rank a single GPU allows
            active = True

        dp = _OneRank(0, 1, args.dp_mode)
    data = make_blobs(device, N_ROWS, D, seed=1)
    e = em.EncoderMap(p, __import__("numpy").zeros((batch, D), "float32"), read_only=True, device=device, dp=dp)
    eng = e._engine
    eng.set_data(data)                      # 512 MB, resident in HBM before the timed region

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    eng.run_steps(args.warmup)
    barrier()
    t0 = time.perf_counter()
    eng.run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if multi:
        tt = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    losses = eng.losses()
    assert all(l == l and abs(l) != float("inf") for l in losses), "non-finite loss in the benchmark run"

    if rank == 0:
        global_batch = batch if rowblock else B * world
        if rowblock:
            objective = (f"one sketch-map cost over one global batch of {batch} rows (the single-process objective at that batch size); "
                         f"dense layers on {world} runs of 16-row blocks, the upper-triangle tiles of the all-pairs pass dealt round-robin")
            scaling = "strong" if batch == B else "weak"
        else:
            objective = (f"mean of {world} independent sketch-map costs over {B} rows each (gradients averaged)" if world > 1
                         else f"one sketch-map cost over one batch of {B} rows")
            scaling = "weak"
        rccl = None
        if multi and backend == "nccl":
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
        out = {
            "metric": "train_samples_per_sec", "value": global_batch * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "steps_per_sec": args.steps / dt, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": "C2: EncoderMap training, 1e6x128 Gaussian blobs resident in HBM, "
                                   + (f"one batch of {batch} rows over all GPUs, " if rowblock else "batch 4096 per GPU, ")
                                   + "sketch-map sigmoid loss (Euclidean), sig=(16,12,6,1,2,6)",
                       "global_batch": global_batch, "features": D, "parallelism": f"dp{world}-{args.dp_mode}" if world > 1 else "single",
                       "objective": objective,
                       "ranks_seen": dist.get_world_size() if multi else 1, "backend": backend if multi else None, "rccl_version": rccl,
                       "allreduce": ("captured in the step graph" if graph_allreduce else "eager between the step's graphs") if multi else None,
                       "graph_allreduce_probe": probe,
                       **({"self_test": f"backend {backend}, one device: {one_device}"} if (one_device or backend != "nccl") else {}),
                       "hip_graph": True},
            "loss_first_last": [float(losses[0]), float(losses[-1])],
        }
        out["roofline"], out["roofline_all_pairs"] = kernel_rooflines(eng, max(args.steps, 50))
        if world == 1 and not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        # RCCL writes its version banner to the C-level stdout, which is flushed at exit -- behind this line unless it is
        # flushed first; the JSON line must be the last thing on stdout
        try:
            import ctypes

            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
