"""The data-parallel step STRUCTURE on one GPU: an RCCL process group of one rank, the engine forced onto its
data-parallel path (two hipGraphs around an eager all_reduce, slice reduction, epilogue behind the optimizer).  The
collective is a one-rank no-op here, so the number is the single-process step plus the structural overhead of the
data-parallel path; the xGMI all-reduce of 8 ranks comes on top.   python tools/bench_dp_structure.py"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import encodermap_amd as em
from encodermap_amd.engine import DataParallel, SequentialEngine


class OneRankDP(DataParallel):
    @property
    def active(self):
        return True


B, D = 4096, 128
rng = np.random.default_rng(1)
cen = rng.normal(0, 4, (16, D))
data = (cen[rng.integers(0, 16, 200000)] + rng.normal(0, 1, (200000, D))).astype(np.float32)
for label, dp in (("single-process path", None), ("data-parallel path, one rank", OneRankDP(0, 1, "replica"))):
    p = em.Parameters(n_steps=2000, periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), seed=3)
    eng = SequentialEngine(D, p, with_distance=True, device="cuda:0", dp=dp)
    eng.set_data(data)
    eng.run_steps(50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run_steps(1000)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 1000
    print(f"{label}: {dt * 1e6:.1f} us/step, loss {eng.losses()[-1]:.4f}", flush=True)
dist.destroy_process_group()
