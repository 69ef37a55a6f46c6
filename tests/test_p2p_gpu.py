"""The one-shot all-reduce over peer-mapped memory (include/emx.h: emx_p2p_*; csrc/p2p.hip) with two and three ranks -- separate
processes that share the test box's one GPU, their regions mapped into one another through the HIP IPC interface (on a multi-GPU
node the same code reads over xGMI).  Every rank must end with the SAME bits: the sum over the ranks in rank order, times the
scale -- which numpy reproduces exactly.  Replaces the RCCL all-reduce of the data-parallel step (the
reference is single-process: autoencoder.py:741-748).  Needs an MI355X."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, n_max, calls):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from encodermap_amd import _lib

    _lib.require_gpu()

    def exchange(h):
        got = [None] * world
        dist.all_gather_object(got, h)
        return got

    ar = _lib.P2PAllReduce(world, rank, n_max, exchange, chunks=8)
    rng = np.random.default_rng(100 + rank)
    outs = []
    ins = []
    import time

    for k, (n, scale) in enumerate(calls):
        x = (rng.normal(0, 1, n) * 10.0 ** rng.integers(-3, 4)).astype(np.float32)
        ins.append(x)
        t = torch.from_numpy(x).cuda()
        if rng.integers(0, 3) == 0:
            time.sleep(float(rng.uniform(0, 0.02)))      # ranks out of step: a launch may wait for its peers, or find them two epochs on
        ar(t, scale)
        outs.append(t)                                   # (no synchronisation between the calls: the launches queue up)
    torch.cuda.synchronize()
    assert not ar.gave_up()
    # a captured graph of four all-reduces on one buffer, replayed three times (the engine captures the collective in its step graph)
    buf = torch.from_numpy(ins[0][:1000].copy()).cuda()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ar(buf, 1.0)                                     # (warm-up on the side stream, as torch asks)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            ar(buf, 0.5)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert not ar.gave_up()
    np.savez(os.path.join(out_dir, f"p2p_{rank}.npz"), graph=buf.cpu().numpy(), first=ins[0][:1000],
             **{f"in{k}": v for k, v in enumerate(ins)}, **{f"out{k}": v.cpu().numpy() for k, v in enumerate(outs)})
    ar.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_p2p_allreduce_between_processes(world, tmp_path):
    import torch.multiprocessing as mp

    calls = [(66698, 1.0 / world), (1, 1.0), (7, 2.0), (4096, 1.0), (66698, 1.0 / world), (33333, 1.0), (66698, 1.0), (5, 1.0)] * 3
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), 66698, calls), nprocs=world, join=True)
    res = [np.load(tmp_path / f"p2p_{r}.npz") for r in range(world)]
    for k, (n, scale) in enumerate(calls):
        want = res[0][f"in{k}"].copy()
        for r in range(1, world):
            want = want + res[r][f"in{k}"]                # float32 additions in rank order, as the kernel's
        want = want * np.float32(scale)
        for r in range(world):
            assert np.array_equal(res[r][f"out{k}"], want), (k, n, r)
    # the graph: 1 warm-up call (scale 1) + 12 captured calls (scale 0.5), each a sum over the ranks of the previous result
    cur = [res[r]["first"].copy() for r in range(world)]
    for it in range(13):
        s = cur[0].copy()
        for r in range(1, world):
            s = s + cur[r]
        s = s * np.float32(1.0 if it == 0 else 0.5)
        cur = [s.copy() for _ in range(world)]
    for r in range(world):
        assert np.array_equal(res[r]["graph"], cur[r]), r
