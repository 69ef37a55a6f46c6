"""Data-parallel host logic on CPU with the gloo backend, world_size 2 and 3.

The HIP kernels cannot run here, so the per-rank compute is played by the float64 oracle; what is under test is the
product's partition + collective plumbing (``encodermap_amd.engine.DataParallel``): runs of 16-row blocks for the dense
layers, the tiles of the symmetric all-pairs pass dealt round-robin, the SUM all-reduce of the [B, L] latent-gradient
shares, the all-reduce of the flat [gradient | loss terms] buffer (SUM for the exact row-block objective, the mean for
independent replica batches).  The row-block scheme must reproduce the single-process gradient and loss exactly."""
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


BATCH = 72     # five 16-row blocks, the last one ragged: 3 + 2 blocks at two ranks, 2 + 2 + 1 (8 rows) at three
TILE = 16      # 5 x 5 tile grid, 15 tiles in the upper triangle: 8 + 7 at two ranks, 5 each at three


def _setup():
    from encodermap_amd.layout import build_layout, init_parameters, unpack_weights
    from encodermap_amd.parameters import Parameters

    rng = np.random.default_rng(0)
    p = Parameters(batch_size=BATCH, n_neurons=[16, 2], activation_functions=["", "tanh", ""])
    x = rng.uniform(-PI, PI, (BATCH, 5)).astype(np.float32)
    lay = build_layout(p.n_neurons, p.activation_functions, 10)
    w0 = unpack_weights(lay, init_parameters(lay, 1))
    return p, x, w0


def _rank_flat(rank, world, mode, p, x, w0, reduce_latent=True, collectives=2):
    """[grad | auto, center, distance] as rank `rank` computes it, following the engine's scheme (engine.py:
    _step_pre_a, _mid_collective, _step_pre_b): encoder on all rows; this rank's dealt tiles of the symmetric all-pairs
    pass (oracle: direct_ops) give its share of d distance / dz of EVERY row, summed over the ranks by the first
    collective; decoder, auto cost, center cost and the backward pass on its own rows."""
    from encodermap_amd.engine import DataParallel

    dp = DataParallel(rank, world, mode, collectives=collectives if mode == "rowblock" else None)
    B = x.shape[0]
    xr = np.roll(x, 7 * rank, axis=0).copy() if mode == "replica" else x     # replica: own mini-batch per rank
    rb, re = dp.row_block(B)
    model = R.SequentialOracle(5, p.n_neurons, p.activation_functions, p.periodicity, w0)
    X = torch.tensor(xr, dtype=torch.float64)
    z = model.encoder(X)
    y = model.decoder(z[rb:re])
    auto = torch.sum(torch.abs(R.periodic_distance(X[rb:re], y, p.periodicity))) / (B * 5) * p.auto_cost_scale
    if mode == "rowblock" and dp.one_collective(B, 5, True):
        # ONE collective (SURVEY.md section 8e as written): the rectangle (own rows) x (all rows) -- the rank's share of the cost and,
        # by the symmetry of the pair terms, the complete d distance / dz of its own rows; nothing to exchange in front of the backward pass
        loss_d, dZ = D.sketchmap_fwd_bwd(xr, z.detach().numpy(), p.periodicity, p.dist_sig_parameters, p.distance_cost_scale, rb, re)
        dZ = torch.tensor(dZ)
    else:
        loss_d, dZ = D.sketchmap_tiles_fwd_bwd(xr, z.detach().numpy(), p.periodicity, p.dist_sig_parameters, p.distance_cost_scale,
                                               dp.dealt_tiles(B, TILE), TILE)
        dZ = torch.tensor(dZ)
        if reduce_latent and mode == "rowblock":     # (replica ranks hold whole batches of their own: no exchange here)
            dp.all_reduce_(dZ, op="sum")
    center = torch.sum(z[rb:re] ** 2) / (B * 2) * p.center_cost_scale
    surrogate = auto + center + torch.sum(dZ[rb:re] * z[rb:re])
    grads = torch.autograd.grad(surrogate, model.params)
    flat = torch.cat([g.reshape(-1) for g in grads] + [torch.tensor([float(auto.detach()), float(center.detach()), loss_d])])
    return dp, flat.detach()


def _worker(rank, world, port, mode, out_dir, collectives=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, x, w0 = _setup()
    dp, flat = _rank_flat(rank, world, mode, p, x, w0, collectives=collectives)
    dp.all_reduce_(flat)
    np.save(os.path.join(out_dir, f"{mode}_{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


def test_dealt_tiles_partition_the_upper_triangle():
    from encodermap_amd.engine import DataParallel

    for B, tile, world in ((72, 16, 2), (72, 16, 3), (4096, 64, 8), (300, 32, 3)):
        nb = -(-B // tile)
        deals = [DataParallel(r, world, "rowblock").dealt_tiles(B, tile) for r in range(world)]
        flat = sorted(t for d in deals for t in d)
        assert flat == [(i, j) for i in range(nb) for j in range(i, nb)]          # every tile exactly once
        assert max(map(len, deals)) - min(map(len, deals)) <= 1
    assert DataParallel(0, 1, "rowblock").dealt_tiles(72, 16) == [(i, j) for i in range(5) for j in range(i, 5)]
    # the oracle's tile form over the whole triangle is the row form over the whole batch
    p, x, w0 = _setup()
    z = np.random.default_rng(1).normal(size=(BATCH, 2))
    full = D.sketchmap_fwd_bwd(x, z, p.periodicity, p.dist_sig_parameters, p.distance_cost_scale)
    tiled = D.sketchmap_tiles_fwd_bwd(x, z, p.periodicity, p.dist_sig_parameters, p.distance_cost_scale,
                                      DataParallel(0, 1, "rowblock").dealt_tiles(BATCH, TILE), TILE)
    np.testing.assert_allclose(tiled[0], full[0], rtol=1e-12)
    np.testing.assert_allclose(tiled[1], full[1], rtol=1e-10, atol=1e-14)


def test_the_form_of_the_row_block_step_is_chosen_per_configuration():
    """one collective (the rectangle: 1 / G more pair work per rank) where that extra work costs less than a latency-bound
    collective, two (dealt triangle) for the large all-pairs passes; an explicit choice wins; the same answer on every rank"""
    from encodermap_amd.engine import DataParallel

    for rank in (0, 7):
        dp = DataParallel(rank, 8, "rowblock")
        assert dp.one_collective(4096, 128, False)          # C2: ~2 us of pair work against a ~25 us collective
        assert not dp.one_collective(16384, 256, False)     # C5
        assert not dp.one_collective(8192, 512, True)       # C4
        assert dp.one_collective(256, 10, True)             # C1
    assert DataParallel(0, 8, "rowblock", collectives=2).one_collective(256, 10, True) is False
    assert DataParallel(0, 8, "rowblock", collectives=1).one_collective(16384, 256, False) is True
    with pytest.raises(ValueError):
        DataParallel(0, 2, "rowblock", collectives=3)


@pytest.mark.parametrize("mode,world,collectives", [("rowblock", 2, 2), ("rowblock", 3, 2), ("rowblock", 2, 1), ("rowblock", 3, 1), ("replica", 2, 2)])
def test_ranks(mode, world, collectives, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), collectives), nprocs=world, join=True)
    got = [np.load(tmp_path / f"{mode}_{r}.npy") for r in range(world)]
    for g in got[1:]:
        np.testing.assert_array_equal(got[0], g)                  # every rank holds the same reduced buffer
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

        parts = [_rank_flat(r, world, mode, p, x, w0, reduce_latent=False)[1].numpy() for r in range(world)]
        assert DataParallel(0, world, mode).row_block(BATCH) == (0, BATCH)
        np.testing.assert_allclose(got[0], (parts[0] + parts[1]) / 2, rtol=1e-12)
