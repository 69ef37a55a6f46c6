"""time of the one-shot all-reduce launch itself with the ranks this process group has (run under torch.distributed.run, or alone:
one rank): 200 launches of the C2 buffer from a graph, HIP events"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from encodermap_amd import _lib
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("gloo", rank=rank, world_size=world)
_lib.require_gpu()
def exchange(h):
    got = [None] * world; dist.all_gather_object(got, h); return got
n = 66698
for chunks in (8, 16, 32, 64):
    ar = _lib.P2PAllReduce(world, rank, n, exchange, chunks=chunks)
    x = torch.randn(n, device="cuda")
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ar(x, 1.0 / world)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50):
            ar(x, 1.0 / world)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    if rank == 0:
        print(f"world {world} chunks {chunks}: {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per all-reduce of {n} floats", flush=True)
    ar.close()
dist.barrier(); dist.destroy_process_group()
