"""One process, the row-block data-parallel path with a group of ONE rank (gloo): is run_steps itself slow?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29613")
dist.init_process_group("gloo", rank=0, world_size=1)
import encodermap_amd as em
from encodermap_amd.engine import DataParallel
class OneRank(DataParallel):
    @property
    def active(self): return True
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "rowblock"
rng = np.random.default_rng(0)
data = rng.normal(0, 1, (50000, 128)).astype(np.float32)
p = em.Parameters(periodicity=float("inf"), batch_size=B, dist_sig_parameters=(16.0, 12, 6, 1, 2, 6), n_steps=400, seed=7)
e = em.EncoderMap(p, data, read_only=True, dp=OneRank(0, 1, mode))
eng = e._engine
eng.run_steps(16); torch.cuda.synchronize()
for n in (1, 10, 100):
    t0 = time.perf_counter(); eng.run_steps(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "B", B, "run_steps(%d): %.3f ms/step" % (n, dt / n * 1e3), "graphs", len(eng._graph))
dist.destroy_process_group()
