"""Device engine of the dense (sequential) autoencoder: ONE training step of the reference's
``SequentialModel.train_step`` (encodermap/models/models.py:3367-3401) as a fixed sequence of HIP kernels
(libemx, through ctypes; torch only owns the device memory and the stream), captured once into a hipGraph
and replayed.

What a step does (reference call stack: SURVEY.md section 3.2):

  shuffle, gather rows (+ sin/cos)   emx_permute, emx_next_batch autoencoder.py:845-856, models.py:3347-3351
  Dense layers forward + auto cost   emx_mlp_forward_backward  models.py:3179-3220, loss_functions.py:586-626,
   + decoder backward                 (one launch)              models.py:3353-3359, 3392-3393
  sketch-map distance + center cost  emx_sketchmap_tiles       loss_functions.py:266-296,351-367,422-449
  encoder backward (+ latent join)   emx_mlp_backward          (tape.gradient, models.py:3392-3393)
  weight gradients, scalar epilogue  emx_mlp_wgrad (+ finish)  models.py:3392-3401 (history / counters on the device)
  [data parallel: replica ranks -- one all-reduce of the flat gradient + loss-term buffer; row-block ranks -- the
   symmetric all-pairs tiles dealt round-robin, an all-reduce of the latent-gradient shares in front of the encoder
   backward, then the same all-reduce: DESIGN.md section 6]
  L2 + clip + Adam, regulariser cost emx_adam_step_slices      autoencoder.py:741-743, loss_functions.py:488-506
   + operand-store update + next batch
(every float32 product of the dense layers and of the Euclidean all-pairs pass runs as three f16 matrix instructions on
operands split into two 16-bit planes: DESIGN.md sections 3 and 4)

The reference evaluates the forward pass once per loss term (4x) plus twice more for the latent
(SURVEY.md section 3.2); the forward is deterministic, so a single pass gives identical numbers.

Memory layout: parameters, gradients (+ 8 loss-term slots), Adam m and v are four flat float32 buffers
(``layout.NetLayout``); per-layer kernels/biases are views.  The training set lives in HBM for the whole
run (N x D float32: 0.5 GB for 1e6 x 128, 4 GB for 2e6 x 512 -- 288 GB per GPU make host staging pointless);
mini-batches are drawn by a device-side permutation stream and gathered by index.
"""
from __future__ import annotations

import gc
import math
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .layout import NetLayout, build_layout, init_parameters, l2_coefficients, pack_weights, unpack_weights

N_TERMS = 8
T_AUTO, T_REG, T_CENTER, T_DIST = 0, 1, 2, 3
TERM_NAMES = ("auto", "reg", "center", "distance")
INF = float("inf")


class DataParallel:
    """Partition of one step over ranks.  ``mode``:

    ``"rowblock"``  every rank sees the same global batch of ``batch_size`` rows and owns a contiguous row block of it for
                    the dense layers: the single-process objective exactly.  The all-pairs cost comes in two forms
                    (``collectives``): ONE collective per step -- SURVEY.md section 8e as written: the rank runs the
                    rectangle (its rows) x (all rows) and has d cost / d z of its own rows by the symmetry of the pair
                    terms, 2 / G of the pair work, the only collective is the all-reduce of [gradient | cost terms] -- or TWO --
                    the symmetric pass over the whole batch with its upper-triangle tiles DEALT to the ranks, 1 / G of the
                    pair work each, and a SUM all-reduce of the [B, L] latent-gradient shares in front of the encoder's
                    backward pass.  ``collectives=None`` chooses per configuration (``one_collective``): the rectangle
                    wherever the pair work it adds costs less than a latency-bound collective (C2: ~2 us against ~25),
                    the dealt triangle for the large all-pairs passes (C4 / C5: hundreds of microseconds per step).
    ``"replica"``   every rank draws its own mini-batch of ``batch_size`` rows and gradients are averaged:
                    classic data parallelism (weak scaling), the objective is the mean of the per-rank costs.
    """

    # what one more latency-bound collective costs a step, and what a pair-element of the all-pairs pass costs a GPU (measured on
    # one MI355X: 17 us at 4096 x 4096 x 128 on the f16 matrix cores, 1.30 ms at 8192 x 8192 x 512 on the periodic VALU tiles)
    COLLECTIVE_SECONDS = 25e-6
    PAIR_ELEMENT_SECONDS = {False: 17e-6 / (4096.0 * 4096.0 * 128.0 / 2), True: 1.30e-3 / (8192.0 * 8192.0 * 512.0 / 2)}

    def __init__(self, rank: int = 0, world_size: int = 1, mode: str = "replica", group=None, collectives: Optional[int] = None,
                 p2p: bool = False):
        """``p2p``: the step's collectives as the library's one-shot all-reduce over peer-mapped memory (include/emx.h: emx_p2p_*) instead
        of the process group's -- ranks of ONE node; one launch per collective, the sum taken in rank order on every rank (bit-identical
        results), captured in the step's graph.  The process group is still needed once per buffer size, to exchange the IPC handles."""
        if mode not in ("rowblock", "replica"):
            raise ValueError(f"data-parallel mode {mode!r} not available")
        if collectives not in (None, 1, 2):
            raise ValueError(f"collectives must be None (chosen per configuration), 1 or 2, got {collectives!r}")
        self.rank, self.world_size, self.mode, self.group = int(rank), int(world_size), mode, group
        self.collectives = collectives
        self.p2p = bool(p2p)
        self._p2p_ctx = {}

    def _p2p_for(self, flat: torch.Tensor):
        """the one-shot all-reduce for buffers of this size (made at the buffer's first collective: on every rank alike, outside any
        stream capture -- the engine's warm-up steps run eagerly)"""
        ar = self._p2p_ctx.get(flat.numel())
        if ar is None:
            import torch.distributed as dist

            def exchange(handle: bytes):
                got = [None] * self.world_size
                dist.all_gather_object(got, handle, group=self.group)
                return got

            ar = self._p2p_ctx[flat.numel()] = _lib.P2PAllReduce(self.world_size, self.rank, flat.numel(), exchange)
        return ar

    def p2p_gave_up(self) -> bool:
        return any(ar.gave_up() for ar in self._p2p_ctx.values())

    def one_collective(self, batch_size: int, n_features: int, periodic: bool) -> bool:
        """Row-block ranks: the rectangle form of the all-pairs pass (one collective per step) or the dealt triangle (two)?
        The rectangle costs every rank 1 / G of the full symmetric pass MORE than the triangle does; it wins where that is less than
        a collective.  The same answer on every rank (a function of the configuration alone)."""
        if self.collectives is not None:
            return self.collectives == 1
        extra = self.PAIR_ELEMENT_SECONDS[bool(periodic)] * (0.5 * batch_size * batch_size * n_features) / max(self.world_size, 1)
        return extra < self.COLLECTIVE_SECONDS

    @property
    def active(self) -> bool:
        return self.world_size > 1

    def row_block(self, batch_size: int):
        """Rows of the global batch owned by this rank: balanced contiguous runs of whole 16-row blocks (the granule of the
        dense-layer kernels' weight-gradient fragments, include/emx.h: emx_mlp_aop_bytes); the last rank takes the ragged end."""
        if not self.active or self.mode == "replica":
            return 0, batch_size
        n_blk = (batch_size + 15) // 16
        if n_blk < self.world_size:
            # (the same error on EVERY rank: a rank with an empty range would fail alone and leave the others in the all-reduce)
            raise ValueError(f"row-block data parallelism needs at least one 16-row block per rank: batch_size {batch_size} gives "
                             f"{n_blk} blocks for {self.world_size} ranks (use mode='replica' or fewer ranks)")
        base, rem = divmod(n_blk, self.world_size)
        begin = self.rank * base + min(self.rank, rem)
        end = begin + base + (1 if self.rank < rem else 0)
        return min(16 * begin, batch_size), min(16 * end, batch_size)

    def dealt_tiles(self, batch_size: int, tile: int):
        """Tiles (ti, tj), ti <= tj, of the symmetric all-pairs pass over the whole batch that this rank computes in the
        row-block mode: the upper triangle of the tile grid in row-major order, dealt round-robin (tile t to rank
        t % world_size -- the rule of emx_sketchmap_tiles_dealt, include/emx.h).  A diagonal tile is half the work of an
        off-diagonal one; with hundreds of tiles per rank the deal is balanced to within one tile."""
        nb = (batch_size + tile - 1) // tile
        out, t = [], 0
        for ti in range(nb):
            for tj in range(ti, nb):
                if not self.active or self.mode == "replica" or t % self.world_size == self.rank:
                    out.append((ti, tj))
                t += 1
        return out

    def all_reduce_(self, flat: torch.Tensor, op: Optional[str] = None) -> None:
        """A collective of a step: gradients and loss terms in one flat buffer (replica: averaged; row blocks: summed), or --
        ``op="sum"`` -- the row-block ranks' contributions to the latent gradient."""
        if not self.active:
            return
        if self.p2p and flat.is_cuda:
            self._p2p_for(flat)(flat, 1.0 / self.world_size if (op != "sum" and self.mode == "replica") else 1.0)
            return
        import torch.distributed as dist

        if op == "sum":
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        if self.mode == "replica" and flat.is_cuda and dist.get_backend(self.group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)   # RCCL averages in the collective
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            if self.mode == "replica":
                flat.mul_(1.0 / self.world_size)

    def batch_seed(self, seed: Optional[int]) -> Optional[int]:
        """rowblock: all ranks must draw identical batches; replica: distinct streams per rank."""
        if seed is None:
            seed = 0 if self.active else None
        if seed is None:
            return None
        return seed if self.mode == "rowblock" else seed * 1000003 + self.rank


class _PlainReconstruction:
    """``Parameters.loss`` = "reconstruction_loss" / "mse" (parameters.py:560-566: "will try to train output == input";
    loss_functions.py:541-547: ``reduce_mean(square(y_pred - y_true))``, nothing else -- SequentialModel.train_step,
    models.py:3385-3386, sums the compiled loss functions only, so neither the L2 regulariser nor the center / distance terms take
    part): the engine's parameter view with the auto cost as the plain mean square at scale 1 and every other term switched off."""

    _OVERRIDES = dict(auto_cost_scale=1.0, auto_cost_variant="mean_square", center_cost_scale=None, distance_cost_scale=None, l2_reg_constant=0.0)

    def __init__(self, p):
        object.__setattr__(self, "_p", p)

    def __getattr__(self, name):
        if name in _PlainReconstruction._OVERRIDES:
            return _PlainReconstruction._OVERRIDES[name]
        return getattr(object.__getattribute__(self, "_p"), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_p"), name, value)


class SequentialEngine:
    """Weights, optimizer state, buffers and the captured step of Autoencoder / EncoderMap."""

    TERM_NAMES = TERM_NAMES
    _reg_slot = T_REG

    def __init__(self, input_dim: int, p, *, with_distance: bool, device: str = "cuda:0",
                 weights: Optional[Sequence[np.ndarray]] = None, dp: Optional[DataParallel] = None,
                 use_graph: bool = True, history_capacity: Optional[int] = None):
        self.plain_loss = getattr(p, "loss", "emap_cost") in ("reconstruction_loss", "mse")
        if self.plain_loss:
            p, with_distance = _PlainReconstruction(p), False
        self.with_distance = bool(with_distance)
        self.periodicity = float(p.periodicity)
        self.periodic = self.periodicity < INF
        # the auto cost's differences: wrapped into the period -- or, for the plain reconstruction loss behind PeriodicOutput, NOT
        # (include/emx.h, emx_auto_cost: a negative periodicity scales the atan2 to the period and leaves the differences alone)
        self._cost_periodicity = (-self.periodicity if self.plain_loss else self.periodicity) if self.periodic else INF
        self.D = int(input_dim)
        self._init_common(2 * self.D if self.periodic else self.D, p, device, weights, dp, use_graph, history_capacity)

    def _init_common(self, net_dim: int, p, device, weights, dp, use_graph, history_capacity) -> None:
        """Flat parameter/optimizer buffers, counters, history -- shared by the sequential and the ADC engine."""
        _lib.require_gpu()
        _lib.load()
        self.p = p
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.dp = dp or DataParallel()
        self.net_dim = int(net_dim)
        self.layout: NetLayout = build_layout(p.n_neurons, p.activation_functions, self.net_dim)
        self.L = self.layout.latent_dim
        self.B = int(p.batch_size)
        self.use_graph = bool(use_graph)
        lay, dev = self.layout, self.device
        P = lay.n_params

        flat = pack_weights(lay, weights) if weights is not None else init_parameters(lay, p.seed)
        self.w = torch.from_numpy(flat).to(dev)
        self.g_ext = torch.zeros(P + N_TERMS, dtype=torch.float32, device=dev)   # gradients | loss terms
        self.g = self.g_ext[:P]
        self.terms = self.g_ext[P:]
        self.m = torch.zeros(P, dtype=torch.float32, device=dev)
        self.v = torch.zeros(P, dtype=torch.float32, device=dev)
        self.l2coef = torch.from_numpy(l2_coefficients(lay, p.l2_reg_constant)).to(dev)
        self.W = [self.w[l.w_off: l.w_off + l.n_in * l.n_out].view(l.n_in, l.n_out) for l in lay.layers]
        self.b = [self.w[l.b_off: l.b_off + l.n_out] for l in lay.layers]

        self.step = torch.full((1,), int(p.current_training_step), dtype=torch.int64, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        # (at least 4096 steps of history: growing it means new buffers, i.e. a new capture of the step's graphs -- a model that is
        #  trained on in small increments must not pay that every time)
        self.history_capacity = max(int(history_capacity or max(int(p.n_steps), 1)), 4096)
        self.history = torch.zeros(self.history_capacity, dtype=torch.float32, device=dev)
        self.term_history = torch.zeros(self.history_capacity, N_TERMS, dtype=torch.float32, device=dev)
        self.hstep = torch.zeros(1, dtype=torch.int64, device=dev)  # write position in history

        self.rb, self.re = self.dp.row_block(self.B)
        # (one stream: a fork / join of the two post-forward branches -- all-pairs kernel || auto cost + decoder backward -- over two
        #  streams cost 5-12 us per cross-stream dependency inside a hipGraph, more than the overlap hid; removed in round 6)
        self._alloc_step_buffers(self.B)
        self._sync_derived_state()
        self._graph = None
        self.data = None
        self.idx_block = None
        self._gen = None

    # ------------------------------------------------------------------------------------------
    def _alloc_step_buffers(self, B: int) -> None:
        lay, dev, f32 = self.layout, self.device, torch.float32
        self.xb = torch.empty(B, self.D, dtype=f32, device=dev)
        self.xin = torch.empty(B, self.net_dim, dtype=f32, device=dev) if self.periodic else self.xb
        # activations h[l] and pre-activation gradients delta[l], [B, n_out]; a row-block rank fills the encoder
        # activations for all B rows (the all-pairs cost needs the full latent) and everything else for its own rows
        self.h = [torch.zeros(B, l.n_out, dtype=f32, device=dev) for l in lay.layers]
        self.delta = [torch.zeros(B, l.n_out, dtype=f32, device=dev) for l in lay.layers]
        self.dY = torch.zeros(B, self.net_dim, dtype=f32, device=dev)
        self.dZ = torch.zeros(B, self.L, dtype=f32, device=dev)
        self.dlat = torch.zeros(B, self.L, dtype=f32, device=dev)   # dh at the latent coming down from the decoder
        self.red_ws2 = torch.zeros(2 * _lib.EMX_REDUCE_WS, dtype=f32, device=dev)   # (two halves: emx_mlp_wgrad_adam_local, by step parity)
        self.sk_ws = torch.zeros(max(_lib.sketchmap_ws_bytes(B, self.L) // 4, 16), dtype=f32, device=dev)   # (zero once: the split reduction keeps its tickets there)
        self.red_ws = torch.zeros(_lib.EMX_REDUCE_WS, dtype=f32, device=dev)
        # f16 planes + norms of the batch rows for the all-pairs kernel (Euclidean data only), made beside the gather
        self.split_ws = None
        if self._prefetch and not self.periodic and self.with_distance and self.p.distance_cost_scale is not None:
            self.split_ws = torch.zeros(_lib.sketchmap_split_bytes(B, self.D), dtype=torch.uint8, device=dev)
        # auto cost per row block (zero: no auto cost); at least EMX_REDUCE_WS long for the stand-alone cost kernel's shares
        self.cost_ws = torch.zeros(max(_lib.mlp_row_blocks(self.re - self.rb), _lib.EMX_REDUCE_WS), dtype=f32, device=dev)
        self.plan = _lib.MlpPlan(lay, self.w, self.h, self.delta, train_rows=B)
        # The same network with the rows NOBODY reads left out of HBM (``_lean``): where the decoder's backward pass rides in the
        # forward launch it takes its activation derivatives from the LDS copies, the weight gradients come from the f16 fragments,
        # and the encoder's backward launch hands nothing but fragments on -- the float32 rows of the decoder's activations and of
        # every delta would be written and never read (C2: 12 of the fused launch's 41.6 MB, 4 of the encoder backward's 16.7).
        # A null row pointer makes the launch drop those stores (a zero-sized buffer descriptor).  h / delta keep the rows of
        # the LAST launch that did write them: diagnostics set ``_lean = False`` (tests/accuracy.py).
        self.plan_lean = self.plan
        if self.plan.windows is None:
            self.plan_lean = _lib.MlpPlan(lay, self.w, [t if i < lay.n_enc else None for i, t in enumerate(self.h)], [None] * len(self.delta),
                                          mirror=self.plan.mirror, train_rows=B, aop=self.plan.aop)
        # slice partials of emx_mlp_wgrad, or the workgroups' shares of emx_mlp_wgrad_adam_local (narrow networks: the larger of the two)
        self.wg_ws = torch.zeros(max(_lib.mlp_wgrad_ws_floats(lay.n_params, self.re - self.rb), _lib.mlp_wgrad_adam_local_ws_floats(self.plan, self.re - self.rb)),
                                 dtype=f32, device=dev)
        self._sk = self._describe_sketchmap() if self._use_sketchmap() else None
        # the all-pairs tiles as workgroups of the fused dense-layer launch (emx_mlp_forward_backward_tiles): the latent rows as the
        # chain workgroups publish them, and the launch's counters (zeroed once; the launch leaves them at zero)
        self.zpub = torch.zeros(B, self.L, dtype=f32, device=dev)
        self.sk_sync = torch.zeros(8, dtype=torch.int32, device=dev)
        self._tiles_ran = False
        # counters of the step's last launch (emx_mlp_wgrad_adam_local), zeroed once
        self.wg_sync = torch.zeros(4 + 1024, dtype=torch.int32, device=dev)
        self._local_ok = None
        self._tail_ran = False

    # -- data feed -------------------------------------------------------------------------------
    def set_data(self, data) -> None:
        """Make the training set device-resident (float32 [N, D])."""
        if isinstance(data, np.ndarray):
            data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
        self.data = data.to(self.device, dtype=torch.float32).contiguous()
        if self.data.shape[1] != self.D:
            raise Exception(f"train data has {self.data.shape[1]} features, the model expects {self.D}")
        # periodic data: the span of the set, once (one device reduction) -- sets that span less than 1.5 periods run the
        # all-pairs tiles on fixed-point phases (include/emx.h: data_span of emx_sketchmap_tiles)
        span = None
        if self.periodic and self.data.numel():
            lo, hi = (float(v) for v in torch.aminmax(self.data))
            span = hi - lo if np.isfinite(hi - lo) else None
        # the captured step holds the old set's pointer and row count (emx_next_batch) and, for periodic data, the tile kernel its
        # span selected: a new training set always means a new capture
        self._graph = None
        self._graph_multi = None
        self._data_span = span
        self._init_batch_stream()

    def _init_batch_stream(self) -> None:
        seed = self.dp.batch_seed(self.p.seed)
        if seed is None:
            seed = int(np.random.SeedSequence().entropy & (2 ** 63 - 1))
        self._shuffle_seed, self._shuffle_epoch = int(seed), 0
        self._perm = torch.empty(0, dtype=torch.int64, device=self.device)   # the pass being consumed ...
        self._perm_pos = 0                                                   # ... and how much of it is used up
        self._pending = None
        self._ring_pos = self._ring_filled = 0                               # (the index block starts over: run_steps)
        if self.idx_block is not None:
            self.idx_block.zero_()     # (indices of the OLD set must not be dereferenced by a warm replay that runs ahead of the filled part)

    def draw_index_block(self, n_batches: int) -> torch.Tensor:
        """Device twin of layout.BatchIndexStream: shuffle(N) -> repeat -> batch(B), batches straddle passes.  Every pass is one
        emx_permute (a keyed bijection evaluated per element: no sort, no host work); torch only owns the buffers."""
        if self._pending is not None:      # batches that were drawn ahead of their use and went back to the stream: they come first
            have = self._pending.numel() // self.B
            if n_batches < have:
                first, self._pending = self._pending[: n_batches * self.B], self._pending[n_batches * self.B:]
                return first
            first, self._pending = self._pending, None
            if n_batches == have:
                return first
            return torch.cat([first, self.draw_index_block(n_batches - have)])
        need = n_batches * self.B
        N = self.data.shape[0]
        have = self._perm.numel() - self._perm_pos
        if need <= have:
            # inside the current pass: a view, nothing is copied (the block is consumed -- copied into the index block -- before
            # the next draw; a pass is only ever replaced by a NEW tensor, so the view stays valid)
            out = self._perm[self._perm_pos: self._perm_pos + need]
            self._perm_pos += need
            return out
        # the block straddles passes: the unused tail of the current pass + whole new passes; the last of them stays as the
        # current pass (one emx_permute of N indices per pass, no copy of the remainder per call)
        n_perm = -(-(need - have) // N)
        out = torch.empty(need, dtype=torch.int64, device=self.device)
        out[:have].copy_(self._perm[self._perm_pos:])
        pos = have
        for k in range(n_perm):
            last = k == n_perm - 1
            perm = torch.empty(N, dtype=torch.int64, device=self.device)
            _lib.permute(perm, self._shuffle_seed, self._shuffle_epoch)
            self._shuffle_epoch += 1
            take = min(N, need - pos)
            out[pos: pos + take].copy_(perm[:take])
            pos += take
            if last:
                self._perm, self._perm_pos = perm, take
        return out

    # -- forward / backward ------------------------------------------------------------------------
    _prefetch = True   # the rows of batch t+1 are gathered beside the optimizer of step t (emx_next_batch)
    _fuse_decoder = True   # forward + auto cost + decoder backward in one launch (plan attributes like this one: set by the parity tests)
    _fold_finish = True    # scalar epilogue inside the weight-gradient launch
    _cost_in_forward = True                                          # auto cost inside the forward launch (False: layers too wide)
    _graph_steps = max(1, int(os.environ.get("EMX_GRAPH_STEPS", "32")))  # steps per replay of the largest multi-step graph (1: single-step graph only); 8 -> 32: -0.6 .. -1.0 us per step at C2
    _graph_multi = None
    _data_span = None   # periodic data: max - min over the training set (set_data)
    _dp_graph_allreduce = os.environ.get("EMX_DP_GRAPH_ALLREDUCE", "0") == "1"   # data parallel: capture the all-reduce too (the engine's own default is eager; bench.py sets it after its child-process probe)
    # the all-pairs tiles ride in the fused dense-layer launch, beside the decoder half of it (single process, Euclidean data on
    # split rows, two latent columns; anything else keeps the separate tile launch).
    _overlap_tiles = True
    _overlap_max_rows = 1536   # ... up to this batch size (beyond it the separate launches are faster: profiles/r03_overlap_ab.txt)
    # ... and, where the tiles ride there, the encoder's backward units too (emx_mlp_forward_backward_tiles_join): the dense-layer
    # workgroups wait inside the launch for the tiles' partial sums and carry on down the encoder from their LDS images
    _join_bwd = True
    _bwd_ran = False
    _tail_ran = False      # the step's last launch ran the optimizer too (emx_mlp_wgrad_adam_local)
    # single process, batches of up to 4096 rows, networks of up to 1000 blocks of 32 x 32: the weight gradients WITHOUT slice partials -- a workgroup owns a
    # 32 x 32 block of a kernel over all rows and updates it itself -- with the epilogue and the next batch's rows in the same launch
    # (emx_mlp_wgrad_adam_local: the step's last TWO launches in one, nothing waits for anything)
    _local_wgrad = True
    _local_ok = None
    _tiles_ran = False
    _lean = True   # float32 rows that no launch reads are not written (plan_lean)

    def _gather_current(self) -> None:
        """Rows of the batch under the cursor -> xb / xin (head of an index block; afterwards the optimizer launch of every
        step gathers the next batch)."""
        _lib.gather_rows(self.data, self.idx_block, self.periodicity if self.periodic else INF, self.xb,
                         self.xin if self.periodic else None, cursor=self.cursor, B=self.B, split_ws=self.split_ws)

    def _forward(self) -> bool:
        """Forward pass, auto cost and the decoder's backward pass in ONE launch (two for a row-block rank: encoder on
        every row first).  The activations, the decoder's deltas and the latent gradient of the auto cost (``dlat``) go to
        HBM for the kernels that follow; the cost's per-row-block shares stay in ``cost_ws`` for the step's scalar epilogue.
        Returns False when the decoder's backward pass still has to be launched (no auto cost, a non-linear output layer,
        or layers too wide for the LDS images of the fused launch)."""
        lay, n, p = self.layout, len(self.layout.layers), self.p
        full = (self.rb, self.re) == (0, self.B)
        fuse = p.auto_cost_scale is not None and self._fuse_decoder
        cost = None
        lean = self._lean and self.plan_lean is not self.plan
        pf = self.plan_lean if lean else self.plan        # (the fused launches: rows nobody reads stay on chip)
        if p.auto_cost_scale is not None:
            cost = _lib.make_auto_cost(self.xb, self.D, 0, self.D, self.periodic, self._cost_periodicity, p.auto_cost_variant,
                                       float(p.auto_cost_scale), self.B, (None if lean else self.delta[-1]) if fuse else self.dY, self.cost_ws)
        self._tiles_ran = self._bwd_ran = False
        if fuse:
            if full and self._overlap_tiles and self._tiles_fit():
                if self._join_bwd and self._sk is not None and self._sk.grad.n_slots <= 64:
                    try:      # ... and the encoder's backward units in the same launch: the step's first THREE launches as one
                        _lib.mlp_forward_backward_tiles_join(pf, 0, n, self.xin, 0, self.B, cost, 0, self.xb, p.dist_sig_parameters, self.sk_ws,
                                                             self.split_ws, self.zpub, self.sk_sync, lay.n_enc - 1, self._sk.grad,
                                                             periodicity=self.periodicity if self.periodic else INF, data_span=self._data_span)
                        self._tiles_ran = self._bwd_ran = self._waits_used = True
                        return True
                    except _lib.EmxError as exc:
                        if not exc.unsupported:
                            raise
                        self._join_bwd = False
                try:
                    _lib.mlp_forward_backward_tiles(pf, 0, n, self.xin, 0, self.B, cost, lay.n_enc, self.dlat, self.xb, p.dist_sig_parameters,
                                                    self.sk_ws, self.split_ws, self.zpub, self.sk_sync,
                                                    periodicity=self.periodicity if self.periodic else INF, data_span=self._data_span)
                    self._tiles_ran = self._waits_used = True
                    return True
                except _lib.EmxError as exc:
                    if not exc.unsupported:
                        raise
                    self._overlap_tiles = False    # outside what the tile role is built for: the separate launches from now on
            try:
                if full:
                    _lib.mlp_forward_backward(pf, 0, n, self.xin, 0, self.B, cost, lay.n_enc, dbelow=self.dlat)
                else:
                    _lib.mlp_forward(pf, 0, lay.n_enc, self.xin, 0, self.B)
                    _lib.mlp_forward_backward(pf, lay.n_enc, n, self.h[lay.n_enc - 1], self.rb, self.re, cost, lay.n_enc,
                                              dbelow=self.dlat)
                return True
            except _lib.EmxError as exc:
                if not exc.unsupported:        # only "this plan does not fit" switches the plan; every other error is a bug
                    raise
                self._fuse_decoder = False     # e.g. layers too wide for one LDS image per layer: two launches from now on
                return self._forward()
        if self._cost_in_forward or cost is None:
            try:
                if full:
                    _lib.mlp_forward(self.plan, 0, n, self.xin, 0, self.B, cost=cost)
                else:
                    _lib.mlp_forward(self.plan, 0, lay.n_enc, self.xin, 0, self.B)
                    _lib.mlp_forward(self.plan, lay.n_enc, n, self.h[lay.n_enc - 1], self.rb, self.re, cost=cost)
                return False
            except _lib.EmxError as exc:
                if cost is None or not exc.unsupported:
                    raise
                self._cost_in_forward = False   # output rows too wide to keep their targets in LDS next to the images
        # widest nets: plain forward launch, the auto cost as its own kernel (its workgroup shares land in cost_ws as well)
        if full:
            _lib.mlp_forward(self.plan, 0, n, self.xin, 0, self.B)
        else:
            _lib.mlp_forward(self.plan, 0, lay.n_enc, self.xin, 0, self.B)
            _lib.mlp_forward(self.plan, lay.n_enc, n, self.h[lay.n_enc - 1], self.rb, self.re)
        _lib.periodic_cost_fwd_bwd(self.h[-1], 0, self.D, self.periodic, self.xb, self.D, self._cost_periodicity,
                                   p.auto_cost_variant, float(p.auto_cost_scale), self.rb, self.re, None, self.dY, self.cost_ws, B=self.B)
        return False

    def _use_sketchmap(self) -> bool:
        return self.with_distance and self.p.distance_cost_scale is not None

    def _tiles_fit(self) -> bool:
        """may the all-pairs tiles ride in the fused dense-layer launch?  (the library has the last word: EMX_EUNSUPPORTED)"""
        # Batches of up to 1536 rows only.  Measured on one MI355X (tools/time_overlap.py, tools/time_join_mid.py,
        # profiles/r03_overlap_ab.txt): with the tiles AND the encoder's backward units aboard the step takes 48.3 against 52.3 us at
        # 1024 rows, 52.2 against 55.6 at 1536, 58.4 against 58.1 at 2048, 74 against 64 at 3072, 83 against 73 at 4096.  A chain
        # workgroup needs 128+ registers per lane, so a CU holds ONE tile workgroup (two tiles in flight) beside it where the
        # separate launch keeps eight tiles resident per CU; two tile waves per SIMD run a tile pair in ~7 us (Gram phase 2.4, sigmoid
        # epilogue 2.2, barriers / reductions / stores 2.4), and at 4.25 pairs per CU (4096 rows) that is longer than the decoder
        # half of the chain workgroups it was meant to hide behind.
        if not (self._use_sketchmap() and not self.dp.active and self.L == 2 and self.layout.n_enc >= 1):
            return False
        if self.periodic:      # the 32 x 32 VALU tiles of periodic batches (up to 512 rows: the default configuration's 256)
            return self.B <= min(512, self._overlap_max_rows)
        return self.split_ws is not None and self.B <= self._overlap_max_rows

    def check_sync(self) -> None:
        """Raises if a wait inside one of the merged launches ever gave up (include/emx.h: the sticky word 2 of their counters): the
        tile / dense-layer workgroups of emx_mlp_forward_backward_tiles[_join] (``sk_sync``), the epilogue workgroup of
        emx_mlp_wgrad_adam_local (``wg_sync``).  A give-up never hangs, but the step it
        happened in -- and every step after it -- worked on incomplete sums.  Called wherever a result leaves the engine (losses,
        weights, encodings, checkpoints); one small device read, and only for engines whose step plan contains such a wait."""
        if self.dp.p2p and self.dp.p2p_gave_up():
            raise _lib.EmxError("a rank's one-shot all-reduce (emx_p2p_allreduce) gave up waiting for a peer: the results of this run are "
                                "invalid (DataParallel(p2p=False) reduces through the process group)")
        if not self._waits_used:
            return
        wg, sk = (int(v) for v in torch.stack([self.wg_sync[2], self.sk_sync[2]]).tolist())
        if wg != 0:
            raise _lib.EmxError("a workgroup of the step's last launch (emx_mlp_wgrad_adam_local) gave up waiting for the "
                                "others: the results of this run are invalid (SequentialEngine._local_wgrad = False runs the separate "
                                "launches and report)")
        if sk != 0:
            raise _lib.EmxError("a workgroup of the fused dense-layer launch gave up waiting (tile workgroups for the latent rows, or "
                                "dense-layer workgroups for the tiles' sums): the results of this run are invalid (SequentialEngine._overlap_tiles = "
                                "False runs the separate launches; please report)")

    _waits_used = False   # a launch with an in-launch wait has been issued by this engine (check_sync has something to check)

    def _dealt(self) -> bool:
        """data parallel over row blocks with the all-pairs tiles dealt to the ranks (DataParallel mode "rowblock", two collectives
        per step); False for the rectangle form (one collective: the rank's rows x all rows through the single-process kernels)"""
        return (self.dp.active and self.dp.mode == "rowblock" and self._use_sketchmap()
                and not self.dp.one_collective(self.B, self.D, self.periodic))

    def _latent_costs(self) -> None:
        """all-pairs tile pass of the sketch-map cost (its O(B) reduction is folded into the encoder's backward kernel and
        the scalar epilogue), or the center cost alone.  Row-block data parallelism: this rank's share of the symmetric
        pass over the whole batch, reduced to its contribution to d cost / d z of ALL rows (summed by _mid_collective)."""
        p = self.p
        Z = self.h[self.layout.n_enc - 1]
        if self._dealt():
            cs = p.center_cost_scale
            _lib.sketchmap_tiles_dealt(self.xb, Z, self.periodicity, p.dist_sig_parameters, self.sk_ws, self.split_ws, self.dp.rank,
                                       self.dp.world_size, data_span=self._data_span)
            _lib.sketchmap_reduce_dealt(self.B, self.D, Z, self.periodicity, p.dist_sig_parameters, float(p.distance_cost_scale),
                                        float(cs) if cs is not None else 0.0, self.rb, self.re, self.dp.rank, self.dp.world_size, self.dZ,
                                        self.sk_ws)
        elif self._use_sketchmap():
            _lib.sketchmap_tiles(self.xb, Z, self.periodicity, p.dist_sig_parameters, self.rb, self.re, self.sk_ws, self.split_ws,
                                 data_span=self._data_span)
        else:
            cs = p.center_cost_scale
            _lib.center_cost_fwd_bwd(Z, float(cs) if cs is not None else 0.0, self.rb, self.re, self.terms[T_CENTER:T_CENTER + 1], self.dZ, False)

    def _mid_collective(self) -> None:
        """Row-block data parallelism: the ranks' contributions to the latent gradient are summed before the encoder's
        backward pass (B x L floats: latency-bound)."""
        if self._dealt():
            self.dp.all_reduce_(self.dZ, op="sum")

    def _describe_sketchmap(self):
        p = self.p
        cs = p.center_cost_scale
        rb, re = (0, self.B) if self._dealt() else (self.rb, self.re)   # (dealt tiles: the symmetric layout of the whole batch)
        return _lib.sketchmap_describe(self.xb, self.h[self.layout.n_enc - 1], self.periodicity, p.dist_sig_parameters,
                                       float(p.distance_cost_scale), float(cs) if cs is not None else 0.0, rb, re, self.sk_ws)

    def _jobs_pre(self) -> list:
        """scalar sums that must be final before a data-parallel all-reduce: (src, n, scale, dev_scale, slot, square)"""
        jobs = [(self.cost_ws, self.cost_ws.numel(), 1.0, None, T_AUTO, False)]
        if self._use_sketchmap():
            sk = self._sk
            Z = self.h[self.layout.n_enc - 1]
            jobs.append((sk.tile_loss, sk.n_tiles, sk.loss_scale, None, T_DIST, False))
            jobs.append((Z.data_ptr() + self.rb * self.L * 4, (self.re - self.rb) * self.L, sk.center_loss_scale, None, T_CENTER, True))
        return jobs

    def _step_pre(self) -> None:
        """Everything up to the gradients, single process or replica ranks (no collective inside)."""
        self._step_pre_a()
        self._step_pre_b()

    def _step_pre_a(self) -> None:
        """Forward pass, auto cost, decoder backward, all-pairs tiles.  After the forward pass there are two independent
        branches -- the all-pairs kernel (needs only X and the latent) and the auto cost + decoder backward -- which meet at
        the latent layer; they run back to back on one stream."""
        lay, n = self.layout, len(self.layout.layers)
        decoder_done = self._forward()
        if decoder_done:
            if not self._tiles_ran:      # (else the tiles were workgroups of the forward launch)
                self._latent_costs()
        else:
            _lib.mlp_backward(self.plan, lay.n_enc, n, self.dY, None, 0, self.rb, self.re, dbelow=self.dlat)
            self._latent_costs()

    def _step_pre_b(self) -> None:
        """Encoder backward (the latent gradient joins here) and the weight gradients."""
        lay = self.layout
        if self._bwd_ran:          # (the encoder's backward units ran in the fused launch)
            self._wgrad()
            return
        pb = self.plan_lean if self._lean else self.plan      # (the encoder's deltas leave as fragments only)
        try:
            if self._dealt():
                _lib.mlp_backward(pb, 0, lay.n_enc, self.dlat, self.dZ, lay.n_enc - 1, self.rb, self.re)
            elif self._use_sketchmap():
                _lib.mlp_backward(pb, 0, lay.n_enc, self.dlat, None, lay.n_enc - 1, self.rb, self.re, latent=self._sk.grad)
            else:
                _lib.mlp_backward(pb, 0, lay.n_enc, self.dlat, self.dZ, lay.n_enc - 1, self.rb, self.re)
        except _lib.EmxError as exc:
            if not exc.unsupported or pb is self.plan:
                raise
            # layers too wide for one launch run as segments that hand their float32 rows on in HBM (csrc/wide.h): the plan that
            # keeps them, from now on
            self._lean = False
            return self._step_pre_b()
        self._wgrad()

    def _local_fits(self) -> bool:
        """whether the step ends with emx_mlp_wgrad_adam_local (decided once per batch shape)"""
        if self._local_ok is None:
            self._local_ok = bool(self._local_wgrad and self._prefetch and type(self)._step_post is SequentialEngine._step_post
                                  and self.red_ws2.numel() >= 2 * _lib.EMX_REDUCE_WS
                                  and _lib.mlp_wgrad_adam_local_fits(self.plan, self.re - self.rb, self._next_batch())
                                  and _lib.mlp_wgrad_adam_local_ws_floats(self.plan, self.re - self.rb) <= self.wg_ws.numel())
        return self._local_ok

    def _late_bookkeeping(self) -> bool:
        """Single process: the step's scalar epilogue rides along with the weight-gradient launch (its inputs are complete
        by then: the regulariser shares of the CURRENT weights were left by the previous optimizer launch, or by
        emx_l2_partials at the start), and the optimizer runs last with ``bookkeeping_done``."""
        return not self.dp.active and self._fold_finish

    def _dp_late_bookkeeping(self) -> bool:
        """Data parallel: the rank's cost sums, the regulariser cost and the counters ride along with the weight-gradient
        launch in front of the all-reduce (the terms travel in it); the optimizer launch behind it runs with the bookkeeping
        done and writes the history row of the reduced terms itself (emx_adam_step_history) -- no launch for the epilogue."""
        return self.dp.active and self._fold_finish

    def _reg_job(self):
        # the regulariser cost is the same on every rank: under the SUM all-reduce of the row-block mode each rank adds 1/G of it
        scale = 1.0 / self.dp.world_size if (self.dp.active and self.dp.mode == "rowblock" and self._dp_late_bookkeeping()) else 1.0
        return (self.red_ws2, _lib.EMX_REDUCE_WS, scale, None, self._reg_slot, False)

    def _wgrad(self) -> None:
        """Weight gradients.  Single GPU: the slice partials stay in the workspace and the Adam kernel adds them up
        itself; data parallel: reduced into the flat gradient buffer for the all-reduce."""
        n = len(self.layout.layers)
        fin = None
        self._tail_ran = False
        if self._late_bookkeeping():
            jobs = self._jobs_pre() + [self._reg_job()]
            fin = _lib.make_finish_desc(jobs, self.terms, N_TERMS, self.history, self.term_history, self.hstep, self.cursor, self.step)
            if self._local_fits():
                p = self.p
                _lib.mlp_wgrad_adam_local(self.plan, self.rb, self.re, self.wg_ws, fin, self.m, self.v, self.l2coef, p.learning_rate, 0.9, 0.999, 1e-7, 1.0,
                                          self.red_ws2, self.wg_sync, next_batch=self._next_batch())
                self._tail_ran = self._waits_used = True
                return
        elif self._dp_late_bookkeeping():
            # the rank's shares of the cost terms, summed by the extra workgroup: they travel in the same all-reduce as the
            # gradients; the counters advance here too, the history row follows the all-reduce (inside the optimizer launch)
            fin = _lib.make_finish_desc(self._jobs_pre() + [self._reg_job()], self.terms, N_TERMS, None, None, self.hstep, self.cursor,
                                        self.step)
        _lib.mlp_wgrad(self.plan, 0, n, self.rb, self.re, self.g if self.dp.active else None, self.wg_ws, finish=fin)
        if fin is None and self.dp.active and self._jobs_pre():
            _lib.finish_step(self._jobs_pre(), self.terms, N_TERMS)

    def _step_post(self) -> None:
        """clip + L2 + Adam (+ operand store, + the next batch), and -- unless it rode along with the weight gradients --
        ONE single-workgroup launch for every scalar of the step: cost sums, regulariser cost, history and counters."""
        p = self.p
        if self._tail_ran:          # (the optimizer ran as the tail of the weight-gradient launch)
            return
        reg_job = self._reg_job()
        nxt = self._next_batch() if self._prefetch else None
        if self._late_bookkeeping():
            _lib.adam_step_slices(self.w, self.wg_ws, _lib.mlp_wgrad_slices(self.re - self.rb), self.m, self.v, self.l2coef,
                                  p.learning_rate, 0.9, 0.999, 1e-7, 1.0, self.step, self.red_ws2, plan=self.plan, next_batch=nxt,
                                  bookkeeping_done=True)
            return
        if self._dp_late_bookkeeping():
            _lib.adam_step_history(self.w, self.g, self.m, self.v, self.l2coef, p.learning_rate, 0.9, 0.999, 1e-7, 1.0, self.step,
                                   self.red_ws2, self.terms, N_TERMS, self.history, self.term_history, self.hstep, plan=self.plan,
                                   next_batch=nxt)
            return
        if self.dp.active:
            _lib.adam_step(self.w, self.g, self.m, self.v, self.l2coef, p.learning_rate, 0.9, 0.999, 1e-7, 1.0, self.step, None, self.red_ws2,
                           plan=self.plan, next_batch=nxt)
            jobs = [reg_job]
        else:
            _lib.adam_step_slices(self.w, self.wg_ws, _lib.mlp_wgrad_slices(self.re - self.rb), self.m, self.v, self.l2coef,
                                  p.learning_rate, 0.9, 0.999, 1e-7, 1.0, self.step, self.red_ws2, plan=self.plan, next_batch=nxt)
            jobs = self._jobs_pre() + [reg_job]
        _lib.finish_step(jobs, self.terms, N_TERMS, self.history, self.term_history, self.hstep, self.cursor, self.step)

    def _next_batch(self):
        """what the optimizer launch prepares for the next step on workgroups of its own (emx_next_batch): the rows of its batch"""
        return _lib.make_next_batch(self.data, self.idx_block, self.cursor, self.B, self.periodicity if self.periodic else INF, self.xb,
                                    self.xin if self.periodic else None, self.split_ws)

    def _sync_derived_state(self) -> None:
        """After the weights were written by anything but the optimizer: operand store and regulariser shares."""
        self.plan.refresh_mirror()
        if self._late_bookkeeping() or self._dp_late_bookkeeping():
            _lib.l2_partials(self.w, self.l2coef, self.red_ws2)
            if self.red_ws2.numel() >= 2 * _lib.EMX_REDUCE_WS:      # (both halves: whichever parity the optimizer step has)
                _lib.l2_partials(self.w, self.l2coef, self.red_ws2[_lib.EMX_REDUCE_WS:])

    def _step_body(self) -> None:
        self._step_pre_a()
        self._mid_collective()
        self._step_pre_b()
        self.dp.all_reduce_(self.g_ext)
        self._step_post()

    def _dp_segments(self):
        """The data-parallel step as (launches, collective behind them) pairs: each launch sequence becomes a hipGraph, the
        collectives run eagerly between them."""
        if self._dealt():
            return [(self._step_pre_a, self._mid_collective), (self._step_pre_b, lambda: self.dp.all_reduce_(self.g_ext)), (self._step_post, None)]
        return [(self._step_pre, lambda: self.dp.all_reduce_(self.g_ext)), (self._step_post, None)]

    # -- driver --------------------------------------------------------------------------------------
    def _state_tensors(self):
        return (self.w, self.m, self.v, self.step, self.hstep, self.cursor, self.history, self.term_history)

    def _capture(self) -> None:
        """Warm up on a side stream, then capture the step into hipGraphs (torch.cuda.CUDAGraph).  Single GPU: one
        graph for the whole step.  Data parallel: one graph up to the gradients, the RCCL all-reduce issued eagerly
        on the same stream, one graph for Adam + bookkeeping."""
        self._graph_multi = None
        self._graph_wrap = None
        # the capture replays its largest graph once (below): up to _graph_steps steps from the current position, restored afterwards.
        # Those replays WRITE history rows and READ the index block that far ahead -- make both exist: room in the history here; the
        # index block ends in _block_slack() zeroed slots (run_steps) and is zeroed by set_data, so whatever lies ahead of the cursor
        # is a valid row of the current training set, wherever in the block the cursor stands
        ahead = max(self._multi_sizes(), default=1) + 2
        if int(self.hstep_host()) + ahead > self.history_capacity:
            self._grow_history(int(self.hstep_host()) + ahead)
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        saved = [t.clone() for t in self._state_tensors()]
        with torch.cuda.stream(s):
            if self._prefetch:
                self._gather_current()
            for _ in range(2):
                self._step_body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()

        def restore():
            for t, sv in zip(self._state_tensors(), saved):
                t.copy_(sv)
            self._sync_derived_state()
            if self._prefetch:
                self._gather_current()

        restore()
        # no garbage collection while a stream is capturing: a collected object that owns device resources (an engine of
        # an earlier model with its graphs, an event) would destroy them in the middle of the capture and abort it
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            captured = False
            if self.dp.active and (self._dp_graph_allreduce or self.dp.p2p):   # (the one-shot all-reduce is a plain launch: always captured)
                # EMX_DP_GRAPH_ALLREDUCE=1 (bench.py sets it when its probe passes): the RCCL all-reduce captured inside the
                # step's graph, so that the step is one graph again and replays many at a time.  Measured with a one-rank
                # RCCL group: 92.7 -> 72.4 us per step (profiles/r04_dp_one_rank_*.json).  Library users opt in themselves:
                # it could not be run with more than one GPU here.
                # (thread-local capture mode: the process group's watchdog thread must not invalidate the capture)
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._step_body()
                    gm = {}
                    for k in self._multi_sizes():
                        restore()
                        gk = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gk, pool=g.pool(), capture_error_mode="thread_local"):
                            for _ in range(k):
                                self._step_body()
                        gm[k] = gk
                    self._graph, self._graph_multi, captured = (g,), (gm or None), True
                except Exception as exc:   # fall back to the two-graph form
                    print(f"encodermap_amd: capturing the all-reduce failed ({type(exc).__name__}: {exc}); using the eager all-reduce")
                    self._dp_graph_allreduce = False
                    torch.cuda.synchronize()
                    restore()
            if captured:
                pass
            elif self.dp.active:
                graphs, pool = [], None
                segs = self._dp_segments()
                for launches, _ in segs:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool):
                        launches()
                    pool = pool or g.pool()
                    graphs.append(g)
                self._graph = tuple(graphs)
                # consecutive steps: the optimizer launch of step t and everything of step t + 1 up to its first collective as ONE
                # graph -- a step is then (number of collectives) graph launches instead of one more (each costs ~9 us of idle
                # device between a graph and an eagerly issued collective, profiles/r03_dp_one_rank_timeline.txt)
                restore()
                gw = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gw, pool=pool):
                    segs[-1][0]()
                    segs[0][0]()
                self._graph_wrap = gw
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step_body()
                self._graph = (g,)
                # the same step several times over in one graph: every kernel reads the device-side counters, so replaying
                # it is the same as replaying the single-step graph that often, minus the launch gaps between graphs
                gm = {}
                for k in self._multi_sizes():
                    restore()
                    gk = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gk, pool=g.pool()):
                        for _ in range(k):
                            self._step_body()
                    gm[k] = gk
                self._graph_multi = gm or None
        finally:
            if gc_was_on:
                gc.enable()
        # first replays are slower than later ones (the graph's kernels and nodes are uploaded on first use): pay that here,
        # not inside the first timed multi-step replay of a short warm-up
        restore()
        for g in (*(self._graph_multi or {}).values(), *((self._graph if len(self._graph) == 1 else ()))):
            restore()
            g.replay()
        torch.cuda.synchronize()
        restore()

    def _multi_sizes(self):
        """steps per replay of the multi-step graphs: the largest (EMX_GRAPH_STEPS, default 32) and its halves down to 2, so that
        a short run needs no single-step replays (every replay costs a graph-launch gap of ~10 us on the host side)"""
        out, k = [], self._graph_steps
        while k > 1:
            out.append(k)
            k //= 2
        return out

    _graph_wrap = None   # data parallel: [last segment of step t | first segment of step t + 1] as one graph

    def _replay(self, k: int = 1) -> None:
        """k consecutive steps from the captured graphs"""
        if len(self._graph) == 1:
            for _ in range(k):
                self._graph[0].replay()
            return
        segs = self._dp_segments()
        wrap = self._graph_wrap
        for i in range(k):
            for j, (g, (_, collective)) in enumerate(zip(self._graph, segs)):
                last = j == len(segs) - 1
                if j == 0 and i > 0 and wrap is not None:
                    pass                      # (ran inside the wrap graph behind the previous step's optimizer)
                elif last and i + 1 < k and wrap is not None:
                    wrap.replay()             # this step's optimizer launch + the next step up to its first collective
                else:
                    g.replay()
                if collective is not None:
                    collective()

    def run_steps(self, n_steps: int, index_block: Optional[torch.Tensor] = None) -> None:
        """Enqueue ``n_steps`` training steps (asynchronous).  ``index_block`` ([n_steps*B] int64, device) injects
        the batch composition; by default it is drawn from the shuffle stream."""
        if self.data is None:
            raise Exception("no train data: call set_data() first")
        if n_steps <= 0:
            return
        if int(self.hstep_host()) + n_steps > self.history_capacity:
            self._grow_history(int(self.hstep_host()) + n_steps)
        if index_block is None and not self._append_blocks:
            index_block = self.draw_index_block(n_steps)     # (EMX_APPEND_BLOCKS=0: the drawn block takes the caller's path)
            block, validated = index_block, True
        else:
            block, validated = index_block, False
        if block is not None and block.numel() != n_steps * self.B:
            raise ValueError(f"index block has {block.numel()} entries, expected {n_steps} x {self.B}")
        if index_block is not None and not validated:
            # the gather kernels dereference data + idx * D unchecked: validate a caller's block once, outside the graph
            wrong_dev = block.device.type != self.device.type or (self.device.index is not None and block.device.index != self.device.index)
            if block.dtype != torch.int64 or wrong_dev:
                raise ValueError(f"index block must be an int64 tensor on {self.device}, got {block.dtype} on {block.device}")
            lo, hi = (int(v) for v in torch.aminmax(block))
            if lo < 0 or hi >= self.data.shape[0]:
                raise ValueError(f"index block holds rows in [{lo}, {hi}], the data set has {self.data.shape[0]} rows")
        cap = self._block_capacity()
        # ``cap`` slots take batches; behind them lie zeroed slack slots (row 0: always a valid row) that nothing ever fills: a capture
        # warms up and replays its largest graph from WHEREVER the cursor stands (_capture: up to _graph_steps + 2 batches ahead of it,
        # each launch gathering one batch further), also two slots in front of the end of the block (a history grown at that point)
        slots = cap + self._block_slack()
        if self.idx_block is None or self.idx_block.numel() < slots * self.B:
            self.idx_block = torch.zeros(slots * self.B, dtype=torch.int64, device=self.device)
            self._graph = None
            self._ring_pos = self._ring_filled = 0
        B = self.B
        if index_block is None and self._append_blocks:
            # Batches of the engine's own shuffle stream are APPENDED to the index block behind the cursor, one batch ahead of what
            # this call consumes: the optimizer launch of the last step finds the rows of the next call's first batch, so a call
            # costs one small copy -- no cursor reset, no head gather (three launches and their host time per call: what made a
            # 20-step run 9 % slower per step than a 300-step run).  The block restarts at its front every ~1000 steps.
            done = 0
            while done < n_steps:
                room = cap - 1 - self._ring_pos
                if room <= 0:
                    # restart: the batch under the cursor (drawn, its rows already gathered by the last optimizer launch) and whatever
                    # was drawn behind it move to the front
                    m = self._ring_filled - self._ring_pos
                    self.idx_block[: m * B].copy_(self.idx_block[self._ring_pos * B: self._ring_filled * B].clone())
                    self.cursor.zero_()
                    self._ring_pos, self._ring_filled = 0, m
                    continue
                k = min(room, n_steps - done)
                need = self._ring_pos + k + 1 - self._ring_filled
                if need > 0:
                    fresh = self._ring_filled == 0
                    self.idx_block[self._ring_filled * B: (self._ring_filled + need) * B].copy_(self.draw_index_block(need))
                    self._ring_filled += need
                    if fresh:
                        self.cursor.zero_()
                        if self._prefetch:
                            self._gather_current()
                self._run_k(k)
                self._ring_pos += k
                done += k
            if self._draw_ahead:
                # The NEXT call's batches, drawn behind this call's launches: a call of the same size then starts with its graph
                # launch (the draw and its copy were ~20 us of host time in front of it -- a twentieth of a 20-step call).  The slots
                # lie behind everything the enqueued steps read; the copy is ordered behind them on the stream.
                ahead = min(n_steps, 64, cap - 1 - self._ring_pos) + 1 - (self._ring_filled - self._ring_pos)
                if ahead > 0 and self._ring_filled + ahead <= cap:
                    self.idx_block[self._ring_filled * B: (self._ring_filled + ahead) * B].copy_(self.draw_index_block(ahead))
                    self._ring_filled += ahead
        else:
            # a caller's index block: written at the front of the block, the cursor reset, its first batch gathered.  The batches the
            # shuffle stream had drawn ahead go back to the stream.
            if self._ring_filled > self._ring_pos:
                back = self.idx_block[self._ring_pos * B: self._ring_filled * B].clone()
                self._pending = back if self._pending is None else torch.cat([back, self._pending])
            self._ring_pos = self._ring_filled = 0
            done = 0
            while done < n_steps:
                k = min(cap, n_steps - done)
                self.idx_block[: k * B].copy_(block[done * B: (done + k) * B])
                self.cursor.zero_()
                if self._prefetch:
                    self._gather_current()
                self._run_k(k)
                done += k
        self._hstep_host += n_steps

    def _run_k(self, k: int) -> None:
        """k steps on the batches under the cursor"""
        if self.use_graph:
            if self._graph is None:
                self._capture()
            r = k
            if len(self._graph) == 1 and self._graph_multi:
                for size in sorted(self._graph_multi, reverse=True):   # 8, 4, 2: 20 steps are 8 + 8 + 4, three replays
                    while r >= size:
                        self._graph_multi[size].replay()
                        r -= size
            self._replay(r)
        else:
            for _ in range(k):
                self._step_body()

    _append_blocks = True
    _draw_ahead = True   # the next call's batches drawn behind this call's launches
    _ring_pos = 0        # batches of the index block consumed (= the device cursor)
    _ring_filled = 0     # batches of the index block that hold drawn indices
    _pending = None      # a batch drawn ahead of its use that went back to the stream

    def _block_capacity(self) -> int:
        return 1024

    def _block_slack(self) -> int:
        """zeroed slots behind the index block's capacity (see run_steps): what a capture's warm replays may read ahead of the cursor"""
        return max(self._multi_sizes(), default=1) + 4

    _hstep_host = 0

    def hstep_host(self) -> int:
        return self._hstep_host

    def reserve_history(self, n_steps_ahead: int) -> None:
        """room for ``n_steps_ahead`` more steps of history, made ONCE (a caller that will enqueue them in small portions --
        train() with callbacks due every few steps -- would otherwise grow the buffers, and re-capture the step, repeatedly)"""
        need = int(self.hstep_host()) + int(n_steps_ahead)
        if need > self.history_capacity:
            self._grow_history(need)

    def _grow_history(self, need: int) -> None:
        cap = max(need, 2 * self.history_capacity)
        h = torch.zeros(cap, dtype=torch.float32, device=self.device)
        th = torch.zeros(cap, N_TERMS, dtype=torch.float32, device=self.device)
        h[: self.history_capacity].copy_(self.history)
        th[: self.history_capacity].copy_(self.term_history)
        self.history, self.term_history, self.history_capacity = h, th, cap
        self._graph = None   # captured pointers are stale

    def losses(self, start: int = 0) -> np.ndarray:
        """Total loss per executed step from step ``start`` on (synchronises; only that tail crosses PCIe)."""
        torch.cuda.synchronize(self.device)
        self.check_sync()
        return self.history[start: self._hstep_host].cpu().numpy()

    def last_loss(self) -> float:
        """Loss of the last executed step: ONE float crosses PCIe (what a summary step needs, autoencoder.train)."""
        if self._hstep_host <= 0:
            raise Exception("no step has run yet")
        self.check_sync()
        return float(self.history[self._hstep_host - 1].item())   # (.item() synchronises with the stream)

    def loss_terms(self, start: int = 0) -> dict:
        torch.cuda.synchronize(self.device)
        self.check_sync()
        th = self.term_history[start: self._hstep_host].cpu().numpy()
        return {name: th[:, i] for i, name in enumerate(self.TERM_NAMES)}

    def current_step(self) -> int:
        return int(self.step.item())

    # -- inference -----------------------------------------------------------------------------------
    # encode() / decode() / generate() (autoencoder.py:1110-1195): ONE streaming launch per chunk (emx_mlp_infer: input rows in, result
    # out, sin / cos and atan2 formed on chip).  Networks it does not take (a layer wider than 128, pieces beyond LDS) run the
    # training kernels forward-only, keeping no rows but the last layer's.
    _infer_streaming = os.environ.get("EMX_INFER_STREAM", "1") != "0"

    def _infer(self, x_dev: torch.Tensor, l0: int, l1: int, periodic_in: bool = False, periodic_out: bool = False,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Layers [l0, l1) on an arbitrary number of rows.  ``periodic_in``: ``x_dev`` holds raw angles (PeriodicInput is part of the
        pass); ``periodic_out``: the result is the angles of PeriodicOutput.  ``out``: where the result goes (a device tensor)."""
        lay = self.layout
        n = x_dev.shape[0]
        self.check_sync()      # (the weights this pass reads are a result of the training run)
        width = lay.layers[l1 - 1].n_out // (2 if periodic_out else 1)
        if out is None:
            out = torch.empty(n, width, dtype=torch.float32, device=self.device)
        if n == 0:
            return out
        if self._infer_streaming:
            try:
                _lib.mlp_infer(self.plan, l0, l1, x_dev, out, self.periodicity if periodic_in else 0.0, self.periodicity if periodic_out else 0.0)
                return out
            except _lib.EmxError as err:
                if not err.unsupported:
                    raise
        if periodic_in:
            xin = torch.empty(n, 2 * x_dev.shape[1], dtype=torch.float32, device=self.device)
            _lib.periodic_in(x_dev, self.periodicity, xin)
            x_dev = xin
        last = out if not periodic_out else torch.empty(n, lay.layers[l1 - 1].n_out, dtype=torch.float32, device=self.device)

        def run(keep_all: bool):
            h = [None] * len(lay.layers)
            for i in range(l0, l1):
                if i == l1 - 1:
                    h[i] = last
                elif keep_all:
                    h[i] = torch.empty(n, lay.layers[i].n_out, dtype=torch.float32, device=self.device)
            plan = _lib.MlpPlan(lay, self.w, h, [None] * len(lay.layers), mirror=self.plan.mirror)
            _lib.mlp_forward(plan, l0, l1, x_dev, 0, n, train=False)

        try:
            # only the last layer's rows are written (include/emx.h: NULL entries of h); a network of more than twelve layers runs as
            # windows of launches that hand rows over through HBM: keep them
            run(self.plan.windows is not None)
        except _lib.EmxError as err:
            if not err.unsupported:
                raise
            run(True)          # (a range run in segments hands rows over through HBM at the cuts: keep them)
        if periodic_out:
            D = lay.layers[l1 - 1].n_out // 2
            _lib.periodic_out(last, D, 0, D, self.periodicity, out)
        return out

    @torch.no_grad()
    def encode(self, x: np.ndarray, already_transformed: bool = False, chunk: int = 262144) -> np.ndarray:
        lay = self.layout
        out = np.empty((x.shape[0], self.L), np.float32)
        for s in range(0, x.shape[0], chunk):
            xc = torch.from_numpy(np.ascontiguousarray(x[s: s + chunk], dtype=np.float32)).to(self.device)
            out[s: s + chunk] = self._infer(xc, 0, lay.n_enc, periodic_in=self.periodic and not already_transformed).cpu().numpy()
        return out

    @torch.no_grad()
    def encode_device(self, x_dev: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``encode`` of rows that already live on the device, into a device tensor (no PCIe traffic; what bench.py times)."""
        return self._infer(x_dev, 0, self.layout.n_enc, periodic_in=self.periodic, out=out)

    @torch.no_grad()
    def decode_device(self, z_dev: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``decode`` of latent rows on the device into a device tensor."""
        return self._infer(z_dev, self.layout.n_enc, len(self.layout.layers), periodic_out=self.periodic, out=out)

    @torch.no_grad()
    def encode_resident(self, chunk: int = 1 << 22) -> np.ndarray:
        """``encode`` of the training set from its HBM-resident copy (``set_data``): no host-to-device transfer, only the
        [N, L] result crosses PCIe.  Same kernel, hence the same numbers as ``encode(train_data)``."""
        if self.data is None:
            raise _lib.EmxError("encode_resident: no resident training set (set_data)")
        N = self.data.shape[0]
        out = torch.empty(N, self.L, dtype=torch.float32, device=self.device)
        for s in range(0, N, chunk):
            self.encode_device(self.data[s: s + chunk], out[s: s + chunk])
        return out.cpu().numpy()

    @torch.no_grad()
    def decode_raw(self, z: np.ndarray, chunk: int = 262144) -> np.ndarray:
        """Output of the last Dense layer (before the atan2 of periodic models)."""
        lay = self.layout
        z = np.asarray(z, np.float32)
        out = np.empty((z.shape[0], self.net_dim), np.float32)
        for s in range(0, z.shape[0], chunk):
            zc = torch.from_numpy(np.ascontiguousarray(z[s: s + chunk])).to(self.device)
            out[s: s + chunk] = self._infer(zc, lay.n_enc, len(lay.layers)).cpu().numpy()
        return out

    @torch.no_grad()
    def decode(self, z: np.ndarray, chunk: int = 262144) -> np.ndarray:
        z = np.asarray(z, np.float32)
        out = np.empty((z.shape[0], self.D), np.float32)
        for s in range(0, z.shape[0], chunk):
            zc = torch.from_numpy(np.ascontiguousarray(z[s: s + chunk])).to(self.device)
            out[s: s + chunk] = self.decode_device(zc).cpu().numpy()
        return out

    # -- state ---------------------------------------------------------------------------------------
    def get_weights(self) -> list:
        self.check_sync()
        return unpack_weights(self.layout, self.w.cpu().numpy())

    def set_weights(self, arrays: Sequence[np.ndarray]) -> None:
        self.w.copy_(torch.from_numpy(pack_weights(self.layout, arrays)).to(self.device))
        self._sync_derived_state()

    def state_dict(self) -> dict:
        torch.cuda.synchronize(self.device)
        self.check_sync()
        return {"w": self.w.cpu().numpy(), "m": self.m.cpu().numpy(), "v": self.v.cpu().numpy(),
                "step": np.int64(self.step.item())}

    def load_state_dict(self, sd: dict) -> None:
        for name in ("w", "m", "v"):
            if name in sd:
                getattr(self, name).copy_(torch.from_numpy(np.asarray(sd[name], np.float32)).to(self.device))
        if "w" in sd:
            self._sync_derived_state()
        if "step" in sd:
            self.step.fill_(int(sd["step"]))
