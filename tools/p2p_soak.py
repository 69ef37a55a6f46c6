"""Soak of the one-shot all-reduce between the ranks of this process group (torch.distributed.run --nproc-per-node N; N processes may
share a GPU): K rounds of [random skew, a graph of 16 all-reduces of y <- mean(y) / 2 + (rank + 1)] against the host's recurrence,
every rank bit-identical.   python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 tools/p2p_soak.py [rounds=3000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from encodermap_amd import _lib
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dist.init_process_group("gloo", rank=rank, world_size=world)
_lib.require_gpu()
def exchange(h):
    got = [None] * world; dist.all_gather_object(got, h); return got
n = 66698
ar = _lib.P2PAllReduce(world, rank, n, exchange)
y = torch.full((n,), float(rank + 1), device="cuda")
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    ar(y, 1.0 / world)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(16):
        ar(y, 1.0 / world)
        y.mul_(0.5).add_(float(rank + 1))
y.fill_(float(rank + 1)); torch.cuda.synchronize()
ref = [np.float32(r + 1) for r in range(world)]
rng = np.random.default_rng(rank)
t0 = time.time()
for k in range(rounds):
    if rng.integers(0, 8) == 0:
        time.sleep(float(rng.uniform(0, 0.003)))
    g.replay()
    for _ in range(16):
        m = ref[0]
        for r in range(1, world):
            m = np.float32(m + ref[r])
        m = np.float32(m * np.float32(1.0 / world))
        ref = [np.float32(np.float32(m * np.float32(0.5)) + np.float32(r + 1)) for r in range(world)]
    if k % 500 == 499 or k == rounds - 1:
        torch.cuda.synchronize()
        v = y.cpu().numpy()
        assert (v == v[0]).all() and v[0] == ref[rank], (k, v[0], ref[rank])
assert not ar.gave_up()
if rank == 0:
    print(f"{world} ranks: {rounds * 16} all-reduces in {time.time() - t0:.1f} s, every checkpoint bit-identical to the host recurrence", flush=True)
ar.close(); dist.barrier(); dist.destroy_process_group()
