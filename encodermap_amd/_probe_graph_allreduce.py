"""Probe: does an RCCL all-reduce captured inside a hipGraph work on THIS machine with THIS many ranks?

Run as a child process by ``bench.py`` (one child per rank, before the parent touches the GPU; RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_ADDR come from the launcher, the children meet on their own port).  Exit code 0:
  (1) a graph holding [scale, all_reduce(AVG), add] was captured and replayed 200 times with the right result on every rank, and
  (2) the ENGINE's own data-parallel step (EMX_PROBE_MODE: "replica" -- one all-reduce of [gradient | terms] -- or "rowblock" --
      the latent-gradient all-reduce as well), captured with its collectives inside the step's graphs, ran 40 steps at the
      benchmark's layer widths and batch with finite losses and bit-identical weights on every rank.
Anything else -- an exception, a wrong number, or a hang that the parent ends with its timeout -- means: keep the all-reduce
eager.  A hang costs only this child: its queues die with it.

EMX_PROBE_P2P=1 asks the other question instead: does the library's one-shot all-reduce over peer-mapped memory (emx_p2p_*, csrc/p2p.hip)
work between THESE ranks -- (1) 60 all-reduces of changing data, eagerly and from a replayed graph, every rank's result bit-identical
to every other's and equal to RCCL's to rounding; (2) the engine's data-parallel step with DataParallel(p2p=True), 40 captured steps,
finite losses, identical weights on every rank, the losses equal to the RCCL run's to 1e-4.
"""
import os
import sys


def main() -> int:
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ["MASTER_PORT"] = os.environ["EMX_PROBE_PORT"]
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
    n = 66700 + 8                                   # the flat [gradient | loss terms] buffer of the benchmark model
    if os.environ.get("EMX_PROBE_P2P") == "1":
        ok = p2p_stage(rank, world, n) and engine_stage(rank, world, os.environ.get("EMX_PROBE_MODE", "replica"), p2p=True)
        flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        torch.cuda.synchronize()
        dist.destroy_process_group()
        return 0 if float(flag) == 1.0 else 3
    x = torch.full((n,), float(rank + 1), device="cuda")
    for _ in range(3):                              # communicator set-up and warm-up outside any capture
        dist.all_reduce(x, op=dist.ReduceOp.AVG)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(8):                      # eight "steps" per graph, as the engine replays them
                x.mul_(2.0)
                dist.all_reduce(x, op=dist.ReduceOp.AVG)
                x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    x.fill_(float(rank + 1))
    torch.cuda.synchronize()
    for _ in range(25):
        g.replay()
    torch.cuda.synchronize()
    # (x has overflowed to inf by now -- v <- 2 v + 1, 200 times: that graph only has to replay without hanging.)  The
    # number that is checked comes from a bounded recurrence:  y <- mean_ranks(y) / 2 + (rank + 1)
    y = torch.full((n,), float(rank + 1), device="cuda")
    gy = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gy, capture_error_mode="thread_local"):
            dist.all_reduce(y, op=dist.ReduceOp.AVG)
            y.mul_(0.5).add_(float(rank + 1))
    torch.cuda.current_stream().wait_stream(s)
    y.fill_(float(rank + 1))
    torch.cuda.synchronize()
    ref = [float(r + 1) for r in range(world)]
    for _ in range(200):
        gy.replay()
        m = sum(ref) / world
        ref = [0.5 * m + (r + 1) for r in range(world)]
    torch.cuda.synchronize()
    ok = abs(float(y[0]) - ref[rank]) < 1e-3 * max(1.0, abs(ref[rank])) and bool((y == y[0]).all())
    if ok and os.environ.get("EMX_PROBE_ENGINE", "1") != "0":
        ok = engine_stage(rank, world, os.environ.get("EMX_PROBE_MODE", "replica"))
    flag = torch.tensor([1.0 if ok else 0.0], device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    return 0 if float(flag) == 1.0 else 3


def p2p_stage(rank: int, world: int, n: int) -> bool:
    """the one-shot all-reduce itself against RCCL: changing data, eager launches and a replayed graph"""
    import torch
    import torch.distributed as dist

    from encodermap_amd import _lib

    def exchange(h):
        got = [None] * world
        dist.all_gather_object(got, h)
        return got

    ar = _lib.P2PAllReduce(world, rank, n, exchange)
    g = torch.Generator(device="cuda")
    g.manual_seed(1000 + rank)
    ok = True
    for it in range(40):
        x = torch.randn(n, generator=g, device="cuda") * (10.0 ** (it % 5 - 2))
        want = x.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        ar(x, 1.0)
        lo, hi = x.clone(), x.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        scale = float(want.abs().max()) + 1e-30
        ok = ok and bool((lo == hi).all()) and float((x - want).abs().max()) <= 1e-5 * scale
    # replayed: y <- mean over the ranks of y, then y <- y / 2 + (rank + 1); the host knows the numbers
    y = torch.full((n,), float(rank + 1), device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ar(y, 1.0 / world)                           # (a warm-up launch outside the capture; undone below)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gy = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gy, capture_error_mode="thread_local"):
            ar(y, 1.0 / world)
            y.mul_(0.5).add_(float(rank + 1))
    torch.cuda.current_stream().wait_stream(s)
    y.fill_(float(rank + 1))
    torch.cuda.synchronize()
    ref = [float(r + 1) for r in range(world)]
    for _ in range(20):
        gy.replay()
        m = sum(ref) / world
        ref = [0.5 * m + (r + 1) for r in range(world)]
    torch.cuda.synchronize()
    ok = ok and abs(float(y[0]) - ref[rank]) < 1e-4 * max(1.0, abs(ref[rank])) and bool((y == y[0]).all()) and not ar.gave_up()
    if ok and (world > 1 or os.environ.get("EMX_PROBE_TIME") == "1"):      # (EMX_PROBE_TIME: the one-rank self-test of this block)
        # and it has to be the FASTER collective on this machine: 4 x 50 launches each, from graphs where RCCL lets itself be captured
        # (the question the first probe answers; eagerly otherwise), the slowest rank's time
        def timed(fn, captured):
            z = torch.zeros(n, device="cuda")
            run = lambda: [fn(z) for _ in range(50)]
            if captured:
                gz = torch.cuda.CUDAGraph()
                with torch.cuda.stream(s):
                    fn(z)
                    torch.cuda.synchronize()
                    with torch.cuda.graph(gz, capture_error_mode="thread_local"):
                        run()
                torch.cuda.current_stream().wait_stream(s)
                run = gz.replay
            run()
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / 200.0], device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t)

        t_p2p = timed(lambda z: ar(z, 1.0 / world), True)
        try:
            t_rccl = timed(lambda z: dist.all_reduce(z, op=dist.ReduceOp.AVG), os.environ.get("EMX_PROBE_RCCL_CAPTURED", "1") == "1")
        except Exception:
            t_rccl = timed(lambda z: dist.all_reduce(z, op=dist.ReduceOp.AVG), False)
        sys.stderr.write(f"probe (rank {rank}): one-shot all-reduce {t_p2p * 1e3:.1f} us, RCCL {t_rccl * 1e3:.1f} us per call of {n} floats\n")
        ok = t_p2p < t_rccl
    ar.close()
    return ok


def engine_stage(rank: int, world: int, mode: str, p2p: bool = False) -> bool:
    """the engine's data-parallel step with its collectives captured (bench.py's C2 shape on a small resident set)"""
    import numpy as np
    import torch
    import torch.distributed as dist

    import encodermap_amd as em
    from encodermap_amd.engine import DataParallel, SequentialEngine

    class _Ranks(DataParallel):      # (a one-rank group still takes the data-parallel path: the self-test of a one-GPU box)
        active = True

    SequentialEngine._dp_graph_allreduce = True
    B, D, rows = 4096, 128, 32768
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    data = torch.randn(rows, D, generator=g, device="cuda") * 3.0
    p = em.Parameters(n_steps=40, seed=7, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6))
    e = em.EncoderMap(p, np.zeros((B, D), "float32"), read_only=True, device=f"cuda:{torch.cuda.current_device()}", dp=_Ranks(rank, world, mode, p2p=p2p))
    eng = e._engine
    eng.set_data(data)
    eng.run_steps(40)
    losses = eng.losses()
    if not eng._dp_graph_allreduce or len(eng._graph) != 1:      # (the capture itself fell back to the eager form)
        return False
    if p2p:
        # the same 40 steps through RCCL: another order of the sums, the same training run to rounding
        e2 = em.EncoderMap(p, np.zeros((B, D), "float32"), read_only=True, device=f"cuda:{torch.cuda.current_device()}", dp=_Ranks(rank, world, mode))
        e2._engine.set_data(data)
        e2._engine.run_steps(40)
        l2 = e2._engine.losses()
        if not (np.isfinite(l2).all() and np.allclose(losses, l2, rtol=1e-4, atol=0)):
            return False
    w = eng.w.double()
    lo, hi = w.clone(), w.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    return bool(np.isfinite(losses).all()) and len(losses) == 40 and bool((lo == hi).all())


if __name__ == "__main__":
    try:
        code = main()
    except Exception as exc:   # noqa: BLE001
        sys.stderr.write(f"probe failed: {type(exc).__name__}: {exc}\n")
        code = 2
    sys.exit(code)
