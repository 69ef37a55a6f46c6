#!/usr/bin/env python3
"""Headline benchmark: EncoderMap training throughput at B=4096, D=128 (BASELINE.json config C2).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full training step of ``em.EncoderMap`` (gather a mini-batch, encoder/decoder forward, sketch-map
distance + auto + center + L2 cost, backward, clip + Adam) on synthetic Gaussian blobs that are resident in HBM
before the timed region starts.  One JSON line is printed by rank 0:

  value        whole-job samples/s = N_gpus * 4096 * K / (max over ranks of the time for K steps)
  roofline     the dominant kernel (sk_tile_f16_kernel, the O(B^2 D) all-pairs pass): algorithmic FLOPs per launch
               (2 per pair-element for the Gram tiles) / its average duration, measured here with HIP events
               around K back-to-back launches on live data, against the 157.3 TFLOP/s fp32 matrix peak -- the precision the
               path delivers (the Gram products run as three f16 matrix instructions per float32 product on split rows,
               csrc/split.h; priced against the f16 peak the executed 6 W FLOPs would be a few per cent: the kernel is bound
               by the VALU work of its sigmoid epilogue).  ``traffic`` = HBM bytes per launch from the rocprofv3 PMC
               passes kept under profiles/.
  cpu_baseline the reference-formulation CPU path (oracle/reference_ops.py in float32: Gram-matrix distances with
               B x B temporaries, autograd, per-element clipped Adam) timed on this host for a few steps of the same
               workload.  N=1 only.  kind "port": TensorFlow itself cannot be installed here.

N > 1: weak scaling -- every rank trains on its own mini-batch of 4096 rows (global batch 4096*N) and the flat
gradient buffer is averaged by one RCCL all-reduce per step (engine.DataParallel mode "replica").
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


def kernel_roofline(eng, launches):
    """Average duration of the tile kernel alone on the engine's live batch (HIP events on the launch stream)."""
    import torch

    from encodermap_amd import _lib

    Z = eng.h[eng.layout.n_enc - 1]
    for _ in range(5):
        _lib.sketchmap_tiles(eng.xb, Z, float("inf"), SIG, 0, B, eng.sk_ws, eng.split_ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        _lib.sketchmap_tiles(eng.xb, Z, float("inf"), SIG, 0, B, eng.sk_ws, eng.split_ws)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / launches
    W = D * B * (B - 1) / 2           # pair-elements per launch (i<->j symmetry exploited)
    flops = 2.0 * W                   # Gram form on the f32 matrix cores: one multiply-add per pair-element (SURVEY.md
                                      # section 8d: "the Gram form would be 2 W"; the direct-difference form counts 3 W)
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.isfile(tf):
        try:
            traffic = json.load(open(tf)).get("sk_tile_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "mfma", "achieved": flops / t / 1e12, "peak": FP32_PEAK / 1e12, "unit": "TFLOP/s",
            "frac": flops / t / FP32_PEAK, "traffic": traffic, "kernel": "sk_tile_f16_kernel", "kernel_us": t * 1e6,
            "algorithmic_bytes": 4 * B * (D + 2 * 2 + 2) + 4,
            "note": "Gram tiles as 3 v_mfma_f32_32x32x16_f16 per float32 product on rows split into two f16 planes (float32-accurate), "
                    "VALU sigmoid epilogue; 2 FLOP per pair-element over the upper triangle, priced against the fp32 matrix peak (the "
                    "delivered precision); algorithmic HBM bytes per launch are ~2 MB so the HBM roof does not bind; the kernel is "
                    "VALU-issue bound (epilogue ~40 instructions per pair)"}


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
    if backend != "nccl":
        graph_allreduce = False             # only an RCCL all-reduce can be captured
    elif os.environ.get("EMX_DP_GRAPH_ALLREDUCE") is None:
        graph_allreduce = multi and probe_graph_allreduce(world)
    else:
        graph_allreduce = os.environ.get("EMX_DP_GRAPH_ALLREDUCE") == "1"
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

    p = em.Parameters(periodicity=float("inf"), batch_size=B, dist_sig_parameters=SIG, n_steps=args.steps + args.warmup, seed=7)
    dp = DataParallel(rank, world, args.dp_mode) if world > 1 else None
    if multi and world == 1:
        class _OneRank(DataParallel):      # the data-parallel path with the one rank a single GPU allows
            active = True

        dp = _OneRank(0, 1, args.dp_mode)
    data = make_blobs(device, N_ROWS, D, seed=1)
    e = em.EncoderMap(p, __import__("numpy").zeros((B, D), "float32"), read_only=True, device=device, dp=dp)
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
        global_batch = B * world if args.dp_mode == "replica" else B
        out = {
            "metric": "train_samples_per_sec", "value": global_batch * args.steps / dt, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "steps_per_sec": args.steps / dt, "higher_is_better": True,
            "scaling": "weak" if args.dp_mode == "replica" else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: EncoderMap training, 1e6x128 Gaussian blobs resident in HBM, batch 4096 per GPU, "
                                   "sketch-map sigmoid loss (Euclidean), sig=(16,12,6,1,2,6)",
                       "global_batch": global_batch, "features": D, "parallelism": f"dp{world}-{args.dp_mode}" if world > 1 else "single",
                       "allreduce": ("captured in the step graph" if graph_allreduce else "eager between two graphs") if multi else None,
                       **({"self_test": f"backend {backend}, one device: {one_device}"} if (one_device or backend != "nccl") else {}),
                       "hip_graph": True},
            "loss_first_last": [float(losses[0]), float(losses[-1])],
        }
        out["roofline"] = kernel_roofline(eng, max(args.steps, 50))
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
