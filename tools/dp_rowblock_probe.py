"""Where does a row-block data-parallel step spend its time?  Two ranks on one GPU over gloo (self-test hooks as bench.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist

def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import encodermap_amd as em
    from encodermap_amd.engine import DataParallel
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rng = np.random.default_rng(0)
    data = rng.normal(0, 1, (50000, 128)).astype(np.float32)
    p = em.Parameters(periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), n_steps=400, seed=7)
    e = em.EncoderMap(p, data, read_only=True, dp=DataParallel(rank, world, "rowblock"))
    eng = e._engine
    eng.run_steps(16)
    torch.cuda.synchronize()
    # time the pieces of one replayed step
    import collections
    acc = collections.defaultdict(float)
    segs = eng._dp_segments()
    for it in range(50):
        for g, (_, coll) in zip(eng._graph, segs):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            g.replay(); torch.cuda.synchronize(); t1 = time.perf_counter()
            if coll is not None: coll()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            acc["graph"] += t1 - t0; acc["collective"] += t2 - t1
    if rank == 0:
        print("B", B, "per step (ms): graphs %.3f  collectives %.3f; graphs per step %d" % (acc["graph"] / 50 * 1e3, acc["collective"] / 50 * 1e3, len(eng._graph)))
        t0 = time.perf_counter(); eng.run_steps(100); torch.cuda.synchronize(); print("run_steps: %.3f ms/step" % ((time.perf_counter() - t0) * 10))
    else:
        eng.run_steps(100)
    dist.destroy_process_group()
main()
