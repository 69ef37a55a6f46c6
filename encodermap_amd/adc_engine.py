"""Device engine of ``AngleDihedralCartesianEncoderMap``: one ``ADCFunctionalModel.train_step``
(encodermap/models/models.py:2260-2521, model graph ``gen_functional_model`` :385-1060) as a captured sequence of
HIP kernels.

Inputs per frame (reference contract, autoencoder.py:1613-1621): central_angles [A-2], central_dihedrals [A-3],
central_cartesians [A,3], central_distances [A-1], side_dihedrals [S] (A = 3 x residues backbone atoms).

Step:
  gather the mini-batch (angular features | distances in one block, cartesians in another)
  encoder input  = [sin|cos] blocks of (angles if use_backbone_angles) | dihedrals | (side dihedrals if use_sidechains)
  dense layers   = emx_mlp_forward, the decoder output is split back into the same blocks           models.py:1499-1541
  predictions    = atan2 per block (PeriodicOutput); angles = batch-mean input angles otherwise (MeanAngles)
  back-mapping   = batch-mean bond lengths -> chain_in_plane -> dihedrals_to_cartesian(dih + pi)     layers.py:957-986
  pair distances = flat upper triangle of atoms start:stop:step, input and back-mapped               layers.py:1252-1267
  costs          = dihedral + angle + cartesian (soft start) + distance + cartesian_distance + center + L2 (+ side)
  backward       = cartesian cost -> pair distances -> back-mapping -> joins the angle/dihedral cost gradients before
                   the atan2 backward; both sketch-map costs and the center cost meet at the latent; dense backward,
                   weight gradients, [all-reduce], Adam.
"""
from __future__ import annotations

import math
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .engine import INF, N_TERMS, DataParallel, SequentialEngine

T_DIH, T_ANG, T_CART, T_DIST, T_CDIST, T_CENTER, T_REG, T_SIDE = range(8)
ADC_TERM_NAMES = ("dihedral", "angle", "cartesian", "distance", "cartesian_distance", "center", "reg", "side_dihedral")


def _round4(n: int) -> int:
    return (n + 3) // 4 * 4


class ADCEngine(SequentialEngine):
    # The head of step t + 1 (batch rows, encoder input, soft-start scale, batch means) runs on workgroups of the optimizer launch
    # of step t, as the row gather of the EncoderMap step does (emx_next_batch::adc_head).  (Plan attributes: the parity tests set them.)
    _prefetch = True
    _fused_head = True      # gathers + sin/cos + soft start in one launch
    _batched_costs = True   # the angle-like costs in one launch
    _fuse_frame_default = True   # the per-frame stages of the step as one launch (emx_adc_frame_*)
    _join_latent = True          # the cartesian-distance cost's latent gradient joins the encoder's backward launch

    TERM_NAMES = ADC_TERM_NAMES
    _reg_slot = T_REG

    def _jobs_pre(self) -> list:
        """The row-mean costs leave their workgroup shares in ``cost_ws`` (one row per term); the step's scalar epilogue adds
        them up.  The all-pairs costs write their terms themselves."""
        jobs = [(self.cost_ws_rows[k], _lib.EMX_REDUCE_WS, 1.0, None, slot, False) for k, slot in self._cost_rows]
        sk = self._sk_cart
        if sk is not None:   # (the cartesian distance cost's tile shares and the center cost: see _step_pre)
            Z = self.h[self.layout.n_enc - 1]
            jobs.append((sk.tile_loss, sk.n_tiles, sk.loss_scale, None, T_CDIST, False))
            jobs.append((Z.data_ptr(), self.B * self.L, sk.center_loss_scale, None, T_CENTER, True))
        return jobs

    def _use_sketchmap(self) -> bool:
        return False   # the ADC step reduces its two all-pairs costs with emx_sketchmap_fwd_bwd

    def __init__(self, n_atoms: int, n_side: int, p, *, device: str = "cuda:0", weights: Optional[Sequence[np.ndarray]] = None,
                 dp: Optional[DataParallel] = None, use_graph: bool = True, history_capacity: Optional[int] = None):
        if p.periodicity == INF:
            raise Exception("AngleDihedralCartesianEncoderMap needs a finite periodicity (angles in radians: 2*pi)")
        if p.use_sidechains and not p.use_backbone_angles:
            raise Exception("Cannot train model with central dihedrals and side dihedrals only. Backbone angles are required.")
        if p.reconstruct_sidechains or p.multimer_training is not None:
            raise NotImplementedError("reconstruct_sidechains / multimer training are experimental in the reference and out of scope here")
        if dp is not None and dp.active and dp.mode != "replica":
            raise NotImplementedError("the ADC engine supports data-parallel mode 'replica' (batch statistics are per rank)")
        self.A = int(n_atoms)
        if self.A < 4:
            raise Exception("at least 4 backbone atoms are needed for a dihedral")
        self.S = int(n_side) if p.use_sidechains else 0
        self.n_ang, self.n_dih, self.n_len = self.A - 2, self.A - 3, self.A - 1
        self.use_ang = bool(p.use_backbone_angles)
        self.periodicity = float(p.periodicity)
        self.periodic = True
        self.with_distance = True
        # feature block: [angles | dihedrals | side | distances]; the encoder sees the first blocks that are switched on
        self.off_ang, self.off_dih = 0, self.n_ang
        self.off_side = self.n_ang + self.n_dih
        self.off_len = self.off_side + self.S
        self.F = self.off_len + self.n_len
        parts = []   # (feature column offset, width, column offset of the sin half inside the network input/output)
        col = 0
        for on, off, n in ((self.use_ang, self.off_ang, self.n_ang), (True, self.off_dih, self.n_dih), (self.S > 0, self.off_side, self.S)):
            if on:
                parts.append((off, n, col))
                col += 2 * n
        self.parts = parts
        self.D = self.F
        # atoms entering the pair distances (cartesian_pwd_start:stop:step)
        sel = range(self.A)[slice(p.cartesian_pwd_start, p.cartesian_pwd_stop, p.cartesian_pwd_step)]
        if len(sel) < 2:
            raise Exception("cartesian_pwd_start/stop/step select fewer than 2 atoms")
        self.pw_start, self.pw_step, self.C = sel.start, sel.step, len(sel)
        self.P = self.C * (self.C - 1) // 2
        self.ldp = _round4(self.P)
        self._fuse_frame = self._fuse_frame_default
        self._fuse_frame_bwd = self._fuse_frame
        self._fuse_cost_jobs = self._fuse_frame
        self._init_common(col, p, device, weights, dp, use_graph, history_capacity)

    # ------------------------------------------------------------------------------------------
    def _alloc_step_buffers(self, B: int) -> None:
        lay, dev, f32 = self.layout, self.device, torch.float32
        z = lambda *shape: torch.zeros(*shape, dtype=f32, device=dev)
        self.feat = z(B, self.F)
        self.cart = z(B, self.A, 3)
        self.xin = z(B, self.net_dim)
        self.xb = self.feat
        self.h = [z(B, l.n_out) for l in lay.layers]
        self.delta = [z(B, l.n_out) for l in lay.layers]
        self.dY = z(B, self.net_dim)
        self.dZ, self.dZ2 = z(B, self.L), z(B, self.L)
        self.red_ws2 = torch.zeros(2 * _lib.EMX_REDUCE_WS, dtype=f32, device=dev)   # (two halves: emx_mlp_wgrad_adam_local, by step parity)
        self.wg_sync = torch.zeros(4 + 1024, dtype=torch.int32, device=dev)
        self.sk_sync = torch.zeros(4, dtype=torch.int32, device=dev)
        self._local_ok = None
        # batch means of the whole feature block in ONE launch: the mean bond lengths (BackMapLayer, layers.py:971) and -- without
        # backbone angles in the model -- the mean angles (MeanAngles, layers.py:1153-1161) are views into it
        self.feat_mean = z(self.F)
        self.ang_pred = z(B, self.n_ang) if self.use_ang else self.feat_mean[self.off_ang: self.off_ang + self.n_ang].view(1, self.n_ang)
        self.dih_pred = z(B, self.n_dih)
        self.mean_len = self.feat_mean[self.off_len: self.off_len + self.n_len]
        self.plane, self.xyz = z(B, self.A, 3), z(B, self.A, 3)
        self.dxyz, self.dplane = z(B, self.A, 3), z(B, self.A, 3)
        self.inp_pair, self.out_pair, self.dpair = z(B, self.ldp), z(B, self.ldp), z(B, self.ldp)
        self.ddih_cart, self.dang_cart = z(B, self.n_dih), z(B, self.n_ang)
        self.cart_scale = z(1)
        dmax = max(self.ldp, self.F)
        self.sk_ws = torch.zeros(max(_lib.sketchmap_ws_bytes(B, self.L) // 4, 16), dtype=f32, device=dev)   # (zero once: the split reduction keeps its tickets there)
        self.split_pair = torch.zeros(_lib.sketchmap_split_bytes(B, self.P), dtype=torch.uint8, device=dev)
        self.red_ws = torch.zeros(_lib.EMX_REDUCE_WS, dtype=f32, device=dev)
        # workgroup shares of the row-mean costs, one row per term: cartesian, dihedral, angle, side dihedral
        self.cost_ws_rows = torch.zeros(4, _lib.EMX_REDUCE_WS, dtype=f32, device=dev)
        self._row_of = {T_CART: 0, T_DIH: 1, T_ANG: 2, T_SIDE: 3}
        self._cost_rows = [(0, T_CART), (1, T_DIH), (2, T_ANG)] + ([(3, T_SIDE)] if self.S else [])
        self.wg_ws = torch.zeros(_lib.mlp_wgrad_ws_floats(lay.n_params, B), dtype=f32, device=dev)
        self.plan = _lib.MlpPlan(lay, self.w, self.h, self.delta, train_rows=B)
        # The cartesian distance cost's O(B) reduction -- tile partials -> d cost / d z and the cost itself -- rides along with the
        # dense layers' backward launch (the latent join of emx_mlp_backward) and the step's scalar epilogue, as in the EncoderMap
        # step, instead of a launch of its own (emx_sketchmap_finish).  Not with the second all-pairs cost switched on: it reuses
        # the tile workspace.  EMX_ADC_JOIN_LATENT=0: the stand-alone reduction.
        p = self.p
        self._sk_cart = None
        if p.cartesian_distance_cost_scale is not None and p.distance_cost_scale is None and self._join_latent:
            cs = p.center_cost_scale
            self._sk_cart = _lib.sketchmap_describe(self.inp_pair[:, :self.P], self.h[lay.n_enc - 1], INF, p.cartesian_dist_sig_parameters,
                                                    float(p.cartesian_distance_cost_scale), float(cs) if cs is not None else 0.0, 0, B,
                                                    self.sk_ws)

    # -- data ------------------------------------------------------------------------------------------
    def set_data(self, angles, dihedrals, cartesians, distances, side_dihedrals=None) -> None:
        A = self.A
        angles, dihedrals, distances = (np.asarray(a, np.float32) for a in (angles, dihedrals, distances))
        cartesians = np.asarray(cartesians, np.float32)
        N = dihedrals.shape[0]
        if cartesians.ndim == 2:
            cartesians = cartesians.reshape(N, -1, 3)
        for name, arr, want in (("central_angles", angles, (N, A - 2)), ("central_dihedrals", dihedrals, (N, A - 3)),
                                ("central_cartesians", cartesians, (N, A, 3)), ("central_distances", distances, (N, A - 1))):
            if arr.shape != want:     # shape asserts of the reference: models.py:1385-1389
                raise Exception(f"{name} has shape {arr.shape}, expected {want}")
        blocks = [angles, dihedrals]
        if self.S:
            if side_dihedrals is None or np.asarray(side_dihedrals).shape != (N, self.S):
                raise Exception(f"side_dihedrals of shape {(N, self.S)} are required with use_sidechains=True")
            blocks.append(np.asarray(side_dihedrals, np.float32))
        blocks.append(distances)
        feats = np.concatenate(blocks, axis=1)
        if np.isnan(feats).any() or np.isnan(cartesians).any():
            raise NotImplementedError("input contains NaNs; the sparse ADC model is not part of the HIP engine")
        self.data = torch.from_numpy(np.ascontiguousarray(feats)).to(self.device)
        self.data_cart = torch.from_numpy(np.ascontiguousarray(cartesians.reshape(N, 3 * A))).to(self.device)
        # the captured step and the head record hold the old set's pointers and row count: a new set means a new capture
        self._graph = None
        self._graph_multi = None
        self._head_key = None
        # span of the angular block the periodic distance cost runs on (None: not finite): below 1.5 periods the all-pairs tiles
        # run on fixed-point phases, beyond that (mixed conventions, unwrapped torsions) the reference's float formula
        lo = self.off_ang if self.use_ang else self.off_dih
        hi = self.off_side + self.S if self.use_ang else self.off_dih + self.n_dih
        vmin, vmax = (float(v) for v in torch.aminmax(self.data[:, lo:hi]))
        self._data_span = (vmax - vmin) if np.isfinite(vmax - vmin) else None
        self._init_batch_stream()

    # -- one step ----------------------------------------------------------------------------------------
    def _encoder_input(self, feat: torch.Tensor, xin: torch.Tensor) -> None:
        for off, n, col in self.parts:
            _lib.periodic_in(feat[:, off: off + n], self.periodicity, xin[:, col: col + 2 * n])

    def _backmap(self, Y: torch.Tensor, feat: torch.Tensor, B: int, ang_pred, dih_pred, mean_len, plane, xyz) -> None:
        """PeriodicOutput -> BackMapLayer (layers.py:957-986)."""
        for off, n, col in self.parts:
            if off == self.off_ang and self.use_ang:
                _lib.periodic_out(Y, n, col, col + n, self.periodicity, ang_pred)
            elif off == self.off_dih:
                _lib.periodic_out(Y, n, col, col + n, self.periodicity, dih_pred)
        if feat is self.feat and mean_len.data_ptr() == self.mean_len.data_ptr():
            _lib.colmean(feat, self.feat_mean)
        else:
            if not self.use_ang:
                _lib.colmean(feat[:, self.off_ang: self.off_ang + self.n_ang], ang_pred.view(-1))
            _lib.colmean(feat[:, self.off_len: self.off_len + self.n_len], mean_len)
        _lib.chain_in_plane_fwd(mean_len, _Bcast(ang_pred) if not self.use_ang else ang_pred, plane)
        _lib.dihedrals_to_cartesian_fwd(dih_pred, math.pi, plane, xyz)

    def _soft_start(self):
        p = self.p
        a, b = p.cartesian_cost_scale_soft_start if p.cartesian_cost_scale_soft_start is not None else (None, None)
        c = p.cartesian_cost_scale
        return (-1.0 if (a is None or b is None) else a, -1.0 if (a is None or b is None) else b, 0.0 if c is None else float(c))

    def _gather_current(self) -> None:
        """the head of the step under the cursor as a launch of its own (head of an index block; afterwards the optimizer launch of
        every step does the head of the next one)"""
        _lib.adc_batch_head(self.data, self.data_cart, self.idx_block, self.cursor, self.B, self.feat, self.cart.view(self.B, -1), self.parts,
                            self.periodicity, self.xin, self.step, *self._soft_start(), self.cart_scale,
                            feat_mean=self.feat_mean if self._fuse_frame else None)

    def _next_batch(self):
        key = (self.data.data_ptr(), self.data_cart.data_ptr(), self.idx_block.data_ptr(), self.idx_block.numel(), self.feat.data_ptr())
        if getattr(self, "_head_key", None) != key:
            self._head_record = _lib.adc_head_describe(self.data, self.data_cart, self.B, self.feat, self.cart.view(self.B, -1), self.parts,
                                                       self.periodicity, self.xin, self.step, *self._soft_start(), self.cart_scale,
                                                       self.feat_mean if self._fuse_frame else None, self.idx_block.numel() // self.B)
            self._head_key = key
        return _lib.make_next_adc_head(self._head_record, self.idx_block, self.cursor, self.B, self.F if self._fuse_frame else 0)

    def _cost_jobs(self, Y: torch.Tensor) -> list:
        p, per = self.p, self.periodicity
        # angle-like costs on the decoder output blocks (gradient of the cartesian path joins before the atan2 backward): one
        # launch for all of them (emx_periodic_cost_batch) -- or none: they ride along with the frame backward pass, behind
        # chain_in_plane's backward pass when the model has backbone angles
        jobs = []
        for off, nn, col in self.parts:
            tgt = self.feat[:, off: off + nn]
            if off == self.off_dih:
                sc, var, ref, slot, extra = p.dihedral_cost_scale, p.dihedral_cost_variant, p.dihedral_cost_reference, T_DIH, self.ddih_cart
            elif off == self.off_ang:
                sc, var, ref, slot, extra = p.angle_cost_scale, p.angle_cost_variant, p.angle_cost_reference, T_ANG, self.dang_cart
            else:
                sc, var, ref, slot, extra = p.side_dihedral_cost_scale, p.side_dihedral_cost_variant, p.side_dihedral_cost_reference, T_SIDE, None
            coef = 0.0 if sc is None else float(sc) / float(ref)
            jobs.append(dict(y=Y, off_sin=col, off_cos=col + nn, periodic_output=True, target=tgt, n=nn, periodicity=per, variant=var, coef=coef,
                             dY=self.dY, ws=self.cost_ws_rows[self._row_of[slot]], dpred_add=extra))
        if not self.use_ang:   # mean angles carry no gradient, the cost is still reported (models.py:2383)
            sc = p.angle_cost_scale
            jobs.append(dict(y=_Bcast(self.ang_pred), off_sin=0, off_cos=0, periodic_output=False,
                             target=self.feat[:, self.off_ang: self.off_ang + self.n_ang], n=self.n_ang, periodicity=per,
                             variant=p.angle_cost_variant, coef=0.0 if sc is None else float(sc) / float(p.angle_cost_reference), dY=None,
                             ws=self.cost_ws_rows[self._row_of[T_ANG]]))
        return jobs

    def _frame_both(self, Y: torch.Tensor, jobs: list) -> bool:
        """emx_adc_frame_fwd_bwd: False outside what either half's fused pass is built for (the two launches, or the single stages, follow)"""
        if not (self._fuse_frame and self._fuse_frame_bwd and self._fuse_cost_jobs and self._frame_both_ok):
            return False
        p = self.p
        cols = {off: (col, col + n) for off, n, col in self.parts}
        try:
            _lib.adc_frame_fwd_bwd(Y, self.periodicity, cols[self.off_ang] if self.use_ang else None,
                                   None if self.use_ang else self.feat_mean[self.off_ang: self.off_ang + self.n_ang],
                                   self.ang_pred if self.use_ang else None, cols[self.off_dih], self.dih_pred, self.mean_len, math.pi, self.plane,
                                   self.xyz, self.cart, self.pw_start, self.pw_step, self.C, self.inp_pair, self.out_pair,
                                   self.split_pair if p.cartesian_distance_cost_scale is not None else None, p.cartesian_cost_variant,
                                   1.0 / float(p.cartesian_cost_reference), self.cart_scale, self.cost_ws_rows[0], self.ddih_cart, self.dplane,
                                   jobs=jobs, dangles=self.dang_cart)
            return True
        except _lib.EmxError as exc:
            if not exc.unsupported:
                raise
            self._frame_both_ok = False
            return False

    _frame_both_ok = True

    def _frame_forward(self, Y: torch.Tensor) -> bool:
        """PeriodicOutput, BackMapLayer and both pair-distance blocks of the step as one launch (emx_adc_frame_fwd) behind the
        batch means; False when the frame does not fit the fused pass (or EMX_ADC_FUSE_FRAME=0): the single stages follow."""
        if not self._fuse_frame:
            return False
        cols = {off: (col, col + n) for off, n, col in self.parts}
        if not self._fused_head:   # (the fused head of the step leaves the batch means as well)
            _lib.colmean(self.feat, self.feat_mean)
        try:
            _lib.adc_frame_fwd(Y, self.periodicity, cols[self.off_ang] if self.use_ang else None,
                               None if self.use_ang else self.feat_mean[self.off_ang: self.off_ang + self.n_ang],
                               self.ang_pred if self.use_ang else None, cols[self.off_dih], self.dih_pred, self.mean_len, math.pi, self.plane,
                               self.xyz, self.cart, self.pw_start, self.pw_step, self.C, self.inp_pair, self.out_pair,
                               split_ws=self.split_pair if self.p.cartesian_distance_cost_scale is not None else None)
        except _lib.EmxError as exc:
            if not exc.unsupported:
                raise
            self._fuse_frame = False
            return False
        return True

    def _frame_backward(self, jobs=None) -> int:
        """Cartesian cost, backward of the pair distances and of dihedrals_to_cartesian as one launch (emx_adc_frame_bwd), with the
        angle-like cost ``jobs`` behind it in the same launch when they fit it.  Returns 0 outside the shapes that pass is built
        for (the single stages follow), 1 when the pass ran, 2 when the jobs ran with it."""
        if not self._fuse_frame_bwd:
            return 0
        p = self.p
        for with_jobs in ((True, False) if (jobs and self._fuse_cost_jobs) else (False,)):
            try:
                _lib.adc_frame_bwd(self.out_pair, self.inp_pair, p.cartesian_cost_variant, 1.0 / float(p.cartesian_cost_reference),
                                   self.cart_scale, self.cost_ws_rows[0], self.xyz, self.dih_pred, math.pi, self.plane, self.pw_start,
                                   self.pw_step, self.C, self.ddih_cart, self.dplane, jobs=jobs if with_jobs else None,
                                   mean_lengths=self.mean_len, angles=self.ang_pred if self.use_ang else None, dangles=self.dang_cart)
                return 2 if with_jobs else 1
            except _lib.EmxError as exc:
                if not exc.unsupported:
                    raise
                if with_jobs:
                    self._fuse_cost_jobs = False
                else:
                    self._fuse_frame_bwd = False
        return 0

    # -- training metrics (ADCClashMetric / ADCRMSDMetric, callbacks/metrics.py:470-581) ------------------------------
    METRIC_WEIGHTS = (14.0067, 24.305, 24.305)   # metrics.py:63

    def _metrics(self) -> None:
        """Clashes of the back-mapped batch and its Kabsch RMSD against the input frames, one value per step like a Keras
        metric (p.track_clashes / p.track_RMSD, autoencoder.py:2255-2258); part of the captured step."""
        p = self.p
        clash = bool(getattr(p, "track_clashes", False))
        rmsd = bool(getattr(p, "track_RMSD", False)) and self.A % 3 == 0
        if not (clash or rmsd):
            return
        if getattr(self, "clash_hist", None) is None or self.clash_hist.numel() != self.history_capacity:
            dev, cap = self.device, self.history_capacity
            old = (getattr(self, "clash_hist", None), getattr(self, "rmsd_hist", None))
            self.clash_hist = torch.zeros(cap, dtype=torch.float32, device=dev)
            self.rmsd_hist = torch.zeros(cap, dtype=torch.float32, device=dev)
            if old[0] is not None:
                self.clash_hist[: old[0].numel()].copy_(old[0])
                self.rmsd_hist[: old[1].numel()].copy_(old[1])
            self.clash_counts = torch.zeros(self.B, dtype=torch.int64, device=dev)
            self.rmsd_rows = torch.zeros(self.B, dtype=torch.float32, device=dev)
        if clash:
            _lib.clash_count(self.xyz, 0.1, self.clash_counts)      # distance_unit "nm" (autoencoder.py:2256)
        if rmsd:
            _lib.kabsch_rmsd(self.cart, self.xyz, self.METRIC_WEIGHTS, True, self.rmsd_rows)
        _lib.metric_tick(self.clash_counts if clash else None, self.rmsd_rows if rmsd else None, self.B, self.clash_hist, self.rmsd_hist,
                         self.hstep)

    def metric_history(self) -> dict:
        """{"ADCClashMetric": [...], "ADCRMSDMetric": [...]} for the steps run so far (only the tracked ones)."""
        out = {}
        if getattr(self, "clash_hist", None) is None:
            return out
        torch.cuda.synchronize(self.device)
        k = self._hstep_host
        if getattr(self.p, "track_clashes", False):
            out["ADCClashMetric"] = self.clash_hist[:k].cpu().numpy()
        if getattr(self.p, "track_RMSD", False) and self.A % 3 == 0:
            out["ADCRMSDMetric"] = self.rmsd_hist[:k].cpu().numpy()
        return out

    def _dealt(self) -> bool:
        return False            # (the ADC step runs single-process or as replicas: one collective per step)

    def _step_pre_a(self) -> None:
        self._step_pre()

    def _step_pre_b(self) -> None:
        pass

    def _step_pre(self) -> None:
        p, B, per = self.p, self.B, self.periodicity
        lay, n = self.layout, len(self.layout.layers)
        t = self.terms
        a, b = p.cartesian_cost_scale_soft_start if p.cartesian_cost_scale_soft_start is not None else (None, None)
        c = p.cartesian_cost_scale
        soft = (-1.0 if (a is None or b is None) else a, -1.0 if (a is None or b is None) else b, 0.0 if c is None else float(c))
        if self._prefetch:
            pass   # (done beside the optimizer of the step in front, or by _gather_current at the head of an index block)
        elif self._fused_head:
            # batch rows, encoder input, the soft-start scale of this step and the batch means of the feature columns in one launch
            _lib.adc_batch_head(self.data, self.data_cart, self.idx_block, self.cursor, B, self.feat, self.cart.view(B, -1), self.parts, per,
                                self.xin, self.step, *soft, self.cart_scale, feat_mean=self.feat_mean if self._fuse_frame else None)
        else:
            _lib.gather_rows(self.data, self.idx_block, INF, self.feat, None, cursor=self.cursor, B=B)
            _lib.gather_rows(self.data_cart, self.idx_block, INF, self.cart.view(B, -1), None, cursor=self.cursor, B=B)
            self._encoder_input(self.feat, self.xin)
        _lib.mlp_forward(self.plan, 0, n, self.xin, 0, B)
        Y, Z = self.h[-1], self.h[lay.n_enc - 1]
        jobs = self._cost_jobs(Y)
        batchable = self._batched_costs and len(jobs) <= _lib.EMX_MAX_COST_JOBS
        # both halves of the per-frame work -- back-mapping with the pair distances, then the cartesian cost and the way back, the
        # angle-like costs behind it -- as ONE launch where everything is in place for it (emx_adc_frame_fwd_bwd)
        both = self._frame_both(Y, jobs) if (batchable and (self._prefetch or self._fused_head)) else False
        if not both and not self._frame_forward(Y):
            self._backmap(Y, self.feat, B, self.ang_pred, self.dih_pred, self.mean_len, self.plane, self.xyz)
            _lib.pdist_flat_fwd(self.cart, self.pw_start, self.pw_step, self.C, self.inp_pair)
            _lib.pdist_flat_fwd(self.xyz, self.pw_start, self.pw_step, self.C, self.out_pair)
        self._metrics()

        # cartesian cost with the soft-start scale of this step, and its way back through the back-mapping
        if not self._fused_head:
            _lib.soft_start_scale(self.step, *soft, self.cart_scale)
        done = 2 if both else self._frame_backward(jobs if batchable else None)
        if done == 0:
            _lib.periodic_cost_fwd_bwd(self.out_pair, 0, 0, False, self.inp_pair, self.P, INF, p.cartesian_cost_variant,
                                       1.0 / float(p.cartesian_cost_reference), 0, B, None, self.dpair, self.cost_ws_rows[0],
                                       coef_scale=self.cart_scale)
            _lib.pdist_flat_bwd(self.xyz, self.dpair, self.pw_start, self.pw_step, self.C, self.dxyz, False)
            _lib.dihedrals_to_cartesian_bwd(self.dih_pred, math.pi, self.plane, self.dxyz, self.ddih_cart, self.dplane)
        if self.use_ang and done == 0:   # (the frame backward pass runs chain_in_plane's backward itself)
            _lib.chain_in_plane_bwd(self.mean_len, self.ang_pred, self.dplane, self.dang_cart)
        if done == 2:
            pass
        elif batchable:
            _lib.periodic_cost_batch(jobs, B, 0, B)
        else:
            for j in jobs:
                _lib.periodic_cost_fwd_bwd(j["y"], j["off_sin"], j["off_cos"], j["periodic_output"], j["target"], j["n"], j["periodicity"],
                                           j["variant"], j["coef"], 0, B, None, j["dY"], j["ws"], B=B, dpred_add=j.get("dpred_add"))

        # latent costs: cartesian distance cost (Euclidean, high-d = input pair distances) with the center cost fused,
        # and the periodic distance cost on the angular inputs when it is switched on
        cs = p.center_cost_scale
        cdc = p.cartesian_distance_cost_scale
        if cdc is not None:
            # the high-d "points" are the frames' pair-distance vectors (thousands of features, few rows): Gram products on the
            # f16 matrix cores from the split rows
            if not self._fuse_frame:   # (the fused frame pass of this step has left the split rows)
                _lib.sketchmap_split(self.inp_pair[:, :self.P], self.split_pair)
            _lib.sketchmap_tiles(self.inp_pair[:, :self.P], Z, INF, p.cartesian_dist_sig_parameters, 0, B, self.sk_ws, self.split_pair)
            if self._sk_cart is None:
                _lib.sketchmap_finish(B, self.P, Z, p.cartesian_dist_sig_parameters, float(cdc), float(cs) if cs is not None else 0.0, 0, B,
                                      t[T_CDIST:T_CDIST + 1], t[T_CENTER:T_CENTER + 1], self.dZ, self.sk_ws)
        else:       # (the slot of a cost that is switched off is never written: it stays at its initial 0)
            _lib.center_cost_fwd_bwd(Z, float(cs) if cs is not None else 0.0, 0, B, t[T_CENTER:T_CENTER + 1], self.dZ, False)
        if p.distance_cost_scale is not None:
            lo = self.off_ang if self.use_ang else self.off_dih
            hi = self.off_side + self.S if self.use_ang else self.off_dih + self.n_dih
            _lib.sketchmap_fwd_bwd(self.feat[:, lo:hi], Z, per, p.dist_sig_parameters, float(p.distance_cost_scale), 0.0, 0, B,
                                   t[T_DIST:T_DIST + 1], None, self.dZ2, self.sk_ws,
                                   data_span=self._data_span)   # measured once by set_data (None / wide: the float form)
            self.dZ.add_(self.dZ2)
        if self._sk_cart is not None:
            _lib.mlp_backward(self.plan, 0, n, self.dY, None, lay.n_enc - 1, 0, B, latent=self._sk_cart.grad)
        else:
            _lib.mlp_backward(self.plan, 0, n, self.dY, self.dZ, lay.n_enc - 1, 0, B)
        self._wgrad()

    # -- cost references (train_for_references, autoencoder.py:1816-1937) -----------------------------------------
    @torch.no_grad()
    def reference_costs(self, n_batches: int, mean_ang: np.ndarray, mean_dih: np.ndarray, mean_len: np.ndarray,
                        index_block: Optional[torch.Tensor] = None) -> dict:
        """Costs of the trivial model that always answers with the data-set means, per batch: dihedral cost, angle cost
        and cartesian cost (pair distances of the mean structure vs the batch), all mean_abs with scale 1 / reference 1
        and periodicity 2 pi as a fresh ``ADCParameters()`` defines them."""
        dev, B = self.device, self.B
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32).reshape(1, -1)).to(dev)
        m_ang, m_dih, m_len = f(mean_ang), f(mean_dih), f(mean_len).view(-1)
        plane = torch.empty(1, self.A, 3, dtype=torch.float32, device=dev)
        xyz = torch.empty_like(plane)
        pd_mean = torch.zeros(1, self.ldp, dtype=torch.float32, device=dev)
        _lib.chain_in_plane_fwd(m_len, m_ang, plane)
        _lib.dihedrals_to_cartesian_fwd(m_dih, math.pi, plane, xyz)
        _lib.pdist_flat_fwd(xyz, self.pw_start, self.pw_step, self.C, pd_mean)
        block = index_block if index_block is not None else self.draw_index_block(n_batches)   # (index_block: injected batches, [n_batches * B])
        if block.numel() != n_batches * B:
            raise ValueError(f"index block has {block.numel()} entries, expected {n_batches} x {B}")
        out = torch.zeros(n_batches, 3, dtype=torch.float32, device=dev)
        two_pi = 2 * math.pi
        for i in range(n_batches):
            idx = block[i * B: (i + 1) * B]
            _lib.gather_rows(self.data, idx, INF, self.feat, None)
            _lib.gather_rows(self.data_cart, idx, INF, self.cart.view(B, -1), None)
            _lib.pdist_flat_fwd(self.cart, self.pw_start, self.pw_step, self.C, self.inp_pair)
            _lib.periodic_cost_fwd_bwd(_Bcast(m_dih), 0, 0, False, self.feat[:, self.off_dih: self.off_dih + self.n_dih], self.n_dih, two_pi,
                                       "mean_abs", 1.0, 0, B, out[i, 0:1], None, self.red_ws, B=B)
            _lib.periodic_cost_fwd_bwd(_Bcast(m_ang), 0, 0, False, self.feat[:, self.off_ang: self.off_ang + self.n_ang], self.n_ang, two_pi,
                                       "mean_abs", 1.0, 0, B, out[i, 1:2], None, self.red_ws, B=B)
            _lib.periodic_cost_fwd_bwd(_Bcast(pd_mean), 0, 0, False, self.inp_pair, self.P, INF, "mean_abs", 1.0, 0, B, out[i, 2:3], None,
                                       self.red_ws, B=B)
        o = out.cpu().numpy()
        # the loop above used the step's batch buffers (feat / cart / inp_pair).  Between two train() calls those hold the HEAD of the
        # next step -- rows, encoder input, soft-start scale, batch means -- left there by the last optimizer launch (append mode:
        # run_steps no longer starts with a head launch): put it back
        if self._prefetch and self.idx_block is not None and self._ring_filled > self._ring_pos:
            self._gather_current()
        return {"dihedral_cost": o[:, 0], "angle_cost": o[:, 1], "cartesian_cost": o[:, 2]}

    # -- inference -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, arrays: Sequence[np.ndarray], chunk: int = 131072) -> np.ndarray:
        """arrays: the encoder inputs in order (angles if use_backbone_angles, dihedrals, side dihedrals if use_sidechains)."""
        if len(arrays) != len(self.parts):
            raise Exception(f"the encoder takes {len(self.parts)} arrays, got {len(arrays)}")
        N = arrays[0].shape[0]
        out = np.empty((N, self.L), np.float32)
        for s in range(0, N, chunk):
            xin = torch.empty(min(chunk, N - s), self.net_dim, dtype=torch.float32, device=self.device)
            for (off, n, col), a in zip(self.parts, arrays):
                if a.shape[1] != n:
                    raise Exception(f"The shape of the provided data ({a.shape}) does not match the expected shape (None, {n}).")
                ac = torch.from_numpy(np.ascontiguousarray(a[s: s + chunk], dtype=np.float32)).to(self.device)
                _lib.periodic_in(ac, self.periodicity, xin[:, col: col + 2 * n])
            out[s: s + chunk] = self._infer(xin, 0, self.layout.n_enc).cpu().numpy()
        return out

    @torch.no_grad()
    def decode(self, z: np.ndarray, chunk: int = 131072) -> list:
        """Latent points -> list of angle arrays, one per decoder block (angles?, dihedrals, side dihedrals?)."""
        z = np.asarray(z, np.float32)
        outs = [np.empty((z.shape[0], n), np.float32) for _, n, _ in self.parts]
        for s in range(0, z.shape[0], chunk):
            zc = torch.from_numpy(np.ascontiguousarray(z[s: s + chunk])).to(self.device)
            y = self._infer(zc, self.layout.n_enc, len(self.layout.layers))
            for (off, n, col), o in zip(self.parts, outs):
                t = torch.empty(y.shape[0], n, dtype=torch.float32, device=self.device)
                _lib.periodic_out(y, n, col, col + n, self.periodicity, t)
                o[s: s + chunk] = t.cpu().numpy()
        return outs

    @torch.no_grad()
    def backmap(self, angles: np.ndarray, dihedrals: np.ndarray, mean_lengths: np.ndarray) -> np.ndarray:
        """chain_in_plane + dihedrals_to_cartesian(dih + pi) for arbitrary frames (generate())."""
        n = dihedrals.shape[0]
        dev = self.device
        ang = torch.from_numpy(np.ascontiguousarray(angles, dtype=np.float32)).to(dev)
        dih = torch.from_numpy(np.ascontiguousarray(dihedrals, dtype=np.float32)).to(dev)
        ml = torch.from_numpy(np.ascontiguousarray(mean_lengths, dtype=np.float32).reshape(-1)).to(dev)
        plane = torch.empty(n, self.A, 3, dtype=torch.float32, device=dev)
        xyz = torch.empty_like(plane)
        _lib.chain_in_plane_fwd(ml, ang if ang.shape[0] == n else _Bcast(ang), plane)
        _lib.dihedrals_to_cartesian_fwd(dih, math.pi, plane, xyz)
        return xyz.cpu().numpy()


class _Bcast:
    """Presents a single row as a [B, n] block with row stride 0 (broadcast) to the C ABI."""

    def __init__(self, t: torch.Tensor):
        self.t = t
        self.is_cuda = t.is_cuda
        self.shape = (1, t.shape[-1])

    def stride(self, i):
        return 0 if i == 0 else 1

    def data_ptr(self):
        return self.t.data_ptr()

    def is_contiguous(self):
        return True
