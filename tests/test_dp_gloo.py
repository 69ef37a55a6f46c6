"""Data-parallel host logic on CPU with the gloo backend, world_size 2.

The HIP kernels cannot run here, so the per-rank compute is played by the float64 oracle; what is under test is the
product's partition + collective plumbing (``encodermap_amd.engine.DataParallel``): row blocks, the single
all-reduce of the flat [gradient | loss terms] buffer, SUM for the exact row-block objective and the mean for
independent replica batches.  The row-block scheme must reproduce the single-process gradient and loss exactly."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import direct_ops as D
from oracle import reference_ops as R

PI = np.pi


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup():
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights
    from encodermap_amd.parameters import Parameters

    rng = np.random.default_rng(0)
    p = Parameters(batch_size=48, n_neurons=[16, 2], activation_functions=["", "tanh", ""])
    x = rng.uniform(-PI, PI, (48, 5)).astype(np.float32)
    lay = build_layout(p.n_neurons, p.activation_functions, 10)
    w0 = unpack_weights(lay, init_parameters(lay, 1))
    return p, x, w0


def _rank_flat(rank, world, mode, p, x, w0):
    """[grad | auto, center, distance] as rank `rank` computes it, following the engine's scheme: encoder on all rows,
    d(distance+center)/dz for own rows from the row-block kernel (oracle: direct_ops), decoder/backward on own rows."""
    from encodermap_amd.engine import DataParallel

    dp = DataParallel(rank, world, mode)
    B = x.shape[0]
    if mode == "replica":                       # own mini-batch: rows differ per rank
        xr = np.roll(x, 7 * rank, axis=0).copy()
        rb, re = dp.row_block(B)
    else:
        xr = x
        rb, re = dp.row_block(B)
    model = R.SequentialOracle(5, p.n_neurons, p.activation_functions, p.periodicity, w0)
    X = torch.tensor(xr, dtype=torch.float64)
    z = model.encoder(X)
    y = model.decoder(z[rb:re])
    auto = torch.sum(torch.abs(R.periodic_distance(X[rb:re], y, p.periodicity))) / (B * 5) * p.auto_cost_scale
    loss_d, dZ = D.sketchmap_fwd_bwd(xr, z.detach().numpy(), p.periodicity, p.dist_sig_parameters, p.distance_cost_scale, rb, re)
    center = torch.sum(z[rb:re] ** 2) / (B * 2) * p.center_cost_scale
    surrogate = auto + center + torch.sum(torch.tensor(dZ[rb:re]) * z[rb:re])
    grads = torch.autograd.grad(surrogate, model.params)
    flat = torch.cat([g.reshape(-1) for g in grads] + [torch.tensor([float(auto), float(center), loss_d])])
    return dp, flat.detach()


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, x, w0 = _setup()
    dp, flat = _rank_flat(rank, world, mode, p, x, w0)
    dp.all_reduce_(flat)
    np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rowblock", "replica"])
def test_two_ranks(mode, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"{mode}_{r}.npy") for r in range(world)]
    np.testing.assert_array_equal(got[0], got[1])                 # every rank holds the same reduced buffer
    p, x, w0 = _setup()
    if mode == "rowblock":
        # single-process reference: full-batch loss and gradient (minus the regulariser, which is added after the reduce)
        model = R.SequentialOracle(5, p.n_neurons, p.activation_functions, p.periodicity, w0)
        total, terms = R.encodermap_loss(model, torch.tensor(x, dtype=torch.float64), p.to_dict())
        grads = torch.autograd.grad(total - terms["reg"], model.params)
        want = np.concatenate([g.reshape(-1).numpy() for g in grads] +
                              [[float(terms["auto"]), float(terms["center"]), float(terms["distance"])]])
        np.testing.assert_allclose(got[0], want, rtol=1e-9, atol=1e-12)
    else:
        # replica mode: mean of the per-rank buffers
        from encodermap_amd.engine import DataParallel

        parts = [_rank_flat(r, world, mode, p, x, w0)[1].numpy() for r in range(world)]
        assert DataParallel(0, world, mode).row_block(48) == (0, 48)
        np.testing.assert_allclose(got[0], (parts[0] + parts[1]) / 2, rtol=1e-12)
