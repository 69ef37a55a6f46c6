"""Reference-formulation restatement (PyTorch CPU, dtype-generic) -- TEST INFRASTRUCTURE.

Every function follows the op sequence of the cited reference file (paths are
relative to the upstream repository AG-Peter/encodermap @ 2025-05-23).  Run with
``torch.float64`` tensors this is the oracle; run with ``torch.float32`` it is
the "reference-formulation CPU path" that ``bench.py`` times as ``cpu_baseline``
(kind "port": it is a restatement, not TensorFlow).

Gradients come from torch autograd through exactly these ops, which is what
``tf.GradientTape`` does for the reference (models/models.py:3367-3401).

Parity status: pinned by the numpy/scipy known answers of the reference's own
tests (tests/test_oracle_kat.py); **unpinned** against TensorFlow-executed
trajectories (TensorFlow is not installable here).
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import numpy as np
import torch

PI = math.pi


# --------------------------------------------------------------------------- #
# a1-a4: encodermap/misc/distances.py
# --------------------------------------------------------------------------- #
def sigmoid(sig: float, a: float, b: float) -> Callable:
    """encodermap/misc/distances.py:66-88 -- 1 - (1 + (2^(a/b) - 1)(r/sig)^a)^(-b/a)."""

    def func(r):
        return 1 - (1 + (2 ** (a / b) - 1) * (r / sig) ** a) ** (-b / a)

    return func


def periodic_distance(a: torch.Tensor, b: torch.Tensor, periodicity: float = 2 * PI) -> torch.Tensor:
    """encodermap/misc/distances.py:113-141."""
    d = torch.abs(b - a)
    return torch.minimum(d, periodicity - d)


def pairwise_dist_periodic(positions: torch.Tensor, periodicity: float) -> torch.Tensor:
    """encodermap/misc/distances.py:144-176 -- materialises [B,B,D]."""
    assert positions.dim() == 2
    vecs = periodic_distance(positions.unsqueeze(1), positions.unsqueeze(0), periodicity)
    mask = (vecs == 0.0).to(positions.dtype)
    vecs = vecs + mask * 1e-12
    return torch.sqrt(torch.sum(torch.square(vecs), dim=2)) + 1.0e-12


def pairwise_dist(positions: torch.Tensor, squared: bool = False, flat: bool = False) -> torch.Tensor:
    """encodermap/misc/distances.py:179-255 -- Gram form, d==0 mask, [1,B,B] for 2-D input."""
    if positions.dim() == 2:
        positions = positions.unsqueeze(0)
    dot = torch.matmul(positions, positions.transpose(1, 2))
    sq = torch.diagonal(dot, dim1=1, dim2=2)
    distances = sq.unsqueeze(1) - 2.0 * dot + sq.unsqueeze(2)
    distances = torch.clamp_min(distances, 0.0)
    if flat:
        n = positions.shape[1]
        iu = torch.triu_indices(n, n, offset=1)
        distances = distances[:, iu[0], iu[1]]
    if not squared:
        mask = (distances == 0.0).to(distances.dtype)
        distances = distances + mask * 1e-16
        distances = torch.sqrt(distances)
        distances = distances * (1.0 - mask)
    return distances


# --------------------------------------------------------------------------- #
# a5-a12: encodermap/loss_functions/loss_functions.py
# --------------------------------------------------------------------------- #
def sigmoid_loss(r_h: torch.Tensor, r_l: torch.Tensor, periodicity: float,
                 dist_sig_parameters: Sequence[float]) -> torch.Tensor:
    """loss_functions.py:351-367 -- mean over all B^2 ordered pairs, diagonal included."""
    if periodicity == float("inf"):
        dist_h = pairwise_dist(r_h)
    else:
        dist_h = pairwise_dist_periodic(r_h, periodicity)
    dist_l = pairwise_dist(r_l)
    sig_h = sigmoid(*dist_sig_parameters[:3])(dist_h)
    sig_l = sigmoid(*dist_sig_parameters[3:])(dist_l)
    return torch.mean(torch.square(sig_h - sig_l))


def _variant_cost(delta: torch.Tensor, variant: str) -> torch.Tensor:
    """loss_functions.py:598-613 (and the identical bodies at :677-694, :757-774, :839-856, :1026-1036)."""
    if variant == "mean_square":
        return torch.mean(torch.square(delta))
    if variant == "mean_abs":
        return torch.mean(torch.abs(delta))
    if variant == "mean_norm":
        return torch.mean(torch.linalg.norm(delta, dim=1))
    raise ValueError(f"cost variant {variant} not available")


def auto_cost(y_true, y_pred, periodicity, variant="mean_abs", scale=1.0):
    """loss_functions.py:586-626; scale None -> 0."""
    if scale is None:
        return y_pred.new_zeros(())
    return _variant_cost(periodic_distance(y_true, y_pred, periodicity), variant) * scale


def center_cost(latent, scale=1e-4):
    """loss_functions.py:422-449."""
    if scale is None:
        return latent.new_zeros(())
    return torch.mean(torch.square(latent)) * scale


def regularization_cost(kernels: Sequence[torch.Tensor], l2_reg_constant: float):
    """loss_functions.py:488-506; Keras l2(lambda) = lambda * sum(w^2), kernels only (models.py:3126,3185)."""
    out = kernels[0].new_zeros(())
    for k in kernels:
        out = out + l2_reg_constant * torch.sum(torch.square(k))
    return out


def distance_cost(x, latent, periodicity, dist_sig_parameters, scale=500.0):
    """loss_functions.py:266-296."""
    if scale is None:
        return latent.new_zeros(())
    return sigmoid_loss(x, latent, periodicity, dist_sig_parameters) * scale


def angle_like_cost(y_true, y_pred, periodicity, variant, reference, scale):
    """dihedral/angle/side_dihedral loss: loss_functions.py:671-706, :832-868, :751-788."""
    if scale is None:
        return y_pred.new_zeros(())
    c = _variant_cost(periodic_distance(y_true, y_pred, periodicity), variant)
    return c / reference * scale


def cartesian_cost(inp_pair, out_pair, variant, reference, current_scale):
    """loss_functions.py:1020-1065 (plain difference, no periodicity)."""
    c = _variant_cost(inp_pair - out_pair, variant)
    return c / reference * current_scale


def soft_start_scale(step: int, soft_start, cartesian_cost_scale) -> float:
    """callbacks/callbacks.py:584-606 (IncreaseCartesianCost.calc_current_cartesian_cost_scale)."""
    if cartesian_cost_scale is None:
        return 0.0
    a, b = soft_start if soft_start is not None else (None, None)
    if a is None or b is None:
        return float(cartesian_cost_scale)
    if step < a:
        return 0.0
    if a <= step <= b:
        return float(cartesian_cost_scale) / (b - a) * (step - a)
    return float(cartesian_cost_scale)


# --------------------------------------------------------------------------- #
# a13/a17: model pieces -- models/models.py:3121-3160,3331-3359; layers.py:204-215,1118-1125
# --------------------------------------------------------------------------- #
def periodic_input(x: torch.Tensor, periodicity: float) -> torch.Tensor:
    """models.py:3347-3351 / layers.py:204-215 -- concat(sin, cos) halves."""
    if periodicity != 2 * PI:
        x = x / periodicity * 2 * PI
    return torch.cat([torch.sin(x), torch.cos(x)], dim=1)


def periodic_output(y: torch.Tensor, periodicity: float) -> torch.Tensor:
    """models.py:3353-3359 / layers.py:1118-1125 -- atan2(first half, second half)."""
    n = y.shape[1] // 2
    out = torch.atan2(y[:, :n], y[:, n:])
    if periodicity != 2 * PI:
        out = out / (2 * PI) * periodicity
    return out


def layer_plan(n_neurons: Sequence[int], activation_functions: Sequence[str], input_dim: int):
    """models.py:3129-3160 -- [(units, act)] for encoder and decoder.

    ``input_dim`` is the width the network sees (already doubled for periodic data).
    """
    acts = [a if a != "" else None for a in activation_functions]
    layer_data = list(zip(list(n_neurons) + list(n_neurons[-2::-1]), acts[1:] + acts[-2::-1]))
    layer_data.append((input_dim, None))
    neurons = [u for u, _ in layer_data]
    bottleneck = neurons.index(min(neurons)) + 1
    return layer_data[:bottleneck], layer_data[bottleneck:]


def keras_activation(name: str, x: torch.Tensor) -> torch.Tensor:
    """tf.keras.activations.<name> (Keras 2.15; models.py:3179-3220 passes the parameter's string to Dense): the ones the product
    builds -- tanh and those whose derivative is a function of the output."""
    if name == "tanh":
        return torch.tanh(x)
    if name == "relu":
        return torch.clamp(x, min=0.0)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "elu":                      # alpha = 1
        return torch.where(x > 0, x, torch.expm1(x))
    if name == "softplus":
        return torch.nn.functional.softplus(x, beta=1.0, threshold=1e9)
    if name == "selu":                     # scale 1.0507..., alpha 1.6732... (keras.activations.selu)
        return 1.0507009873554805 * torch.where(x > 0, x, 1.6732632423543772 * torch.expm1(x))
    if name == "softsign":
        return x / (1.0 + torch.abs(x))
    if name == "exponential":
        return torch.exp(x)
    raise ValueError(f"activation {name!r} not restated")


def dense_stack(x: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor],
                acts: Sequence[Optional[str]]) -> torch.Tensor:
    """Keras Dense: act(x @ W + b), W is [in, out]."""
    for W, b, act in zip(weights, biases, acts):
        x = x @ W + b
        if act is not None:
            x = keras_activation(act, x)
    return x


class SequentialOracle:
    """SequentialModel (models/models.py:3099-3401) as plain tensors.

    ``params`` is the flat list [W0, b0, W1, b1, ...] over encoder then decoder layers,
    the same order as Keras ``trainable_variables``.
    """

    def __init__(self, input_dim: int, n_neurons, activation_functions, periodicity: float,
                 params: Sequence[np.ndarray], dtype=torch.float64):
        self.periodicity = float(periodicity)
        self.periodic = self.periodicity < float("inf")
        self.net_dim = input_dim * 2 if self.periodic else input_dim
        enc, dec = layer_plan(n_neurons, activation_functions, self.net_dim)
        self.enc_acts = [a for _, a in enc]
        self.dec_acts = [a for _, a in dec]
        self.n_enc = len(enc)
        self.params = [torch.tensor(np.asarray(p), dtype=dtype).requires_grad_(True) for p in params]
        self.dtype = dtype

    def _wb(self):
        W = self.params[0::2]
        b = self.params[1::2]
        return W, b

    def encoder(self, x):
        W, b = self._wb()
        if self.periodic:
            x = periodic_input(x, self.periodicity)
        return dense_stack(x, W[: self.n_enc], b[: self.n_enc], self.enc_acts)

    def decoder(self, z):
        W, b = self._wb()
        y = dense_stack(z, W[self.n_enc:], b[self.n_enc:], self.dec_acts)
        if self.periodic:
            y = periodic_output(y, self.periodicity)
        return y

    def kernels(self):
        return self.params[0::2]


def encodermap_loss(model: SequentialOracle, x: torch.Tensor, p: dict, with_distance: bool = True):
    """Sum of the loss list of EncoderMap._get_loss (autoencoder/autoencoder.py:1279-1307)
    evaluated as in SequentialModel.train_step (models/models.py:3385-3386).  The reference
    re-runs the forward once per loss; the forward is deterministic, so one pass is identical.
    Returns (total, dict of terms)."""
    z = model.encoder(x)
    y = model.decoder(z)
    terms = {
        "auto": auto_cost(x, y, model.periodicity, p.get("auto_cost_variant", "mean_abs"), p.get("auto_cost_scale", 1)),
        "reg": regularization_cost(model.kernels(), p.get("l2_reg_constant", 0.001)),
        "center": center_cost(z, p.get("center_cost_scale", 1e-4)),
    }
    if with_distance:
        terms["distance"] = distance_cost(x, z, model.periodicity, p.get("dist_sig_parameters", (4.5, 12, 6, 1, 2, 6)),
                                          p.get("distance_cost_scale", 500))
    total = sum(terms.values())
    return total, terms


def reconstruction_loss(model: SequentialOracle, x: torch.Tensor):
    """``Parameters.loss`` = "reconstruction_loss" / "mse" (loss_functions/loss_functions.py:541-547, autoencoder.py:948-958):
    ``reduce_mean(square(model(x) - x))`` and nothing else -- the differences are plain also behind PeriodicOutput, and the
    train step (models.py:3385-3386) adds neither the L2 regulariser nor any other term.  Returns (total, terms)."""
    y = model.decoder(model.encoder(x))
    total = torch.mean(torch.square(y - x))
    zero = torch.zeros((), dtype=x.dtype)
    return total, {"auto": total, "reg": zero, "center": zero}


# --------------------------------------------------------------------------- #
# a15: Keras Adam with clipvalue (Keras 2.13-2.15 optimizer semantics)
# --------------------------------------------------------------------------- #
class KerasAdam:
    def __init__(self, params: Sequence[torch.Tensor], learning_rate=1e-3, beta_1=0.9, beta_2=0.999,
                 epsilon=1e-7, clipvalue: Optional[float] = 1.0):
        self.lr, self.b1, self.b2, self.eps, self.clip = learning_rate, beta_1, beta_2, epsilon, clipvalue
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.iterations = 0

    @torch.no_grad()
    def apply_gradients(self, params, grads):
        t = self.iterations + 1
        alpha = self.lr * math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
        for p, g, m, v in zip(params, grads, self.m, self.v):
            if self.clip is not None:
                g = torch.clamp(g, -self.clip, self.clip)
            m += (g - m) * (1 - self.b1)
            v += (g * g - v) * (1 - self.b2)
            p -= (m * alpha) / (torch.sqrt(v) + self.eps)
        self.iterations = t


def train_steps(model: SequentialOracle, data: np.ndarray, batch_indices: Sequence[np.ndarray], p: dict,
                with_distance: bool = True, clipvalue: Optional[float] = 1.0):
    """Oracle training loop with injected weights (in ``model``) and injected batch indices.
    Returns (list of total loss per step, list of per-term dicts)."""
    opt = KerasAdam(model.params, learning_rate=p.get("learning_rate", 1e-3), clipvalue=clipvalue)
    X = torch.tensor(np.asarray(data), dtype=model.dtype)
    losses, terms_all = [], []
    for idx in batch_indices:
        xb = X[torch.as_tensor(np.asarray(idx), dtype=torch.long)]
        if p.get("loss", "emap_cost") in ("reconstruction_loss", "mse"):
            total, terms = reconstruction_loss(model, xb)
        else:
            total, terms = encodermap_loss(model, xb, p, with_distance)
        grads = torch.autograd.grad(total, model.params, allow_unused=True)
        grads = [g if g is not None else torch.zeros_like(q) for g, q in zip(grads, model.params)]
        opt.apply_gradients(model.params, grads)
        losses.append(float(total.detach()))
        terms_all.append({k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in terms.items()})
    return losses, terms_all


# --------------------------------------------------------------------------- #
# a18-a20: back-mapping -- encodermap_tf1/backmapping.py, encodermap_tf1/misc.py, models/layers.py
# --------------------------------------------------------------------------- #
def straight_tetrahedral_chain(n_atoms=None, bond_lengths=None) -> np.ndarray:
    """encodermap_tf1/backmapping.py:71-94 (float32 table like the reference)."""
    dx = math.cos(70.63 / 180 * PI)
    dy = math.sin(70.63 / 180 * PI)
    if n_atoms and not bond_lengths:
        coordinates = np.zeros((n_atoms, 3), dtype=np.float32)
        indices = np.repeat(np.arange(int(n_atoms / 2) + 1), 2)
        coordinates[:, 0] = indices[1: n_atoms + 1] + dx * indices[0:n_atoms]
        coordinates[:, 1] = dy * indices[0:n_atoms]
    elif (bond_lengths and not n_atoms) or n_atoms == len(bond_lengths) + 1:
        n_bonds = len(bond_lengths)
        n_atoms = n_atoms or n_bonds + 1
        dxs = bond_lengths * np.tile([1, dx], int(n_atoms / 2))[:n_bonds]
        dys = bond_lengths * np.tile([0, dy], int(n_atoms / 2))[:n_bonds]
        coordinates = np.zeros((n_atoms, 3), dtype=np.float32)
        coordinates[1:, 0] = np.cumsum(dxs)
        coordinates[1:, 1] = np.cumsum(dys)
    else:
        raise ValueError("input not compatible")
    return coordinates


def chain_in_plane(lengths: torch.Tensor, angles: torch.Tensor) -> torch.Tensor:
    """encodermap_tf1/backmapping.py:97-119.  lengths [B or 1, A-1], angles [B, A-2] -> [B, A, 3]."""
    B = angles.shape[0]
    if lengths.shape[0] == 1:
        lengths = lengths.expand(B, -1)
    prev_angle = angles.new_zeros(B)
    xs = [angles.new_zeros(B)]
    ys = [angles.new_zeros(B)]
    sign = 1.0
    n = angles.shape[1]
    for i in range(n):
        xs.append(xs[-1] + lengths[:, i] * torch.cos(prev_angle))
        ys.append(ys[-1] + lengths[:, i] * torch.sin(prev_angle) * sign)
        prev_angle = PI - angles[:, i] - prev_angle
        sign *= -1
    xs.append(xs[-1] + lengths[:, n] * torch.cos(prev_angle))
    ys.append(ys[-1] + lengths[:, n] * torch.sin(prev_angle) * sign)
    xs = torch.stack(xs, dim=1)
    ys = torch.stack(ys, dim=1)
    return torch.stack([xs, ys, torch.zeros_like(xs)], dim=2)


def rotation_matrix(axis_unit_vec: torch.Tensor, angle: torch.Tensor) -> torch.Tensor:
    """encodermap_tf1/misc.py:286-304 -- Rodrigues; returned R is applied as row-vector @ R."""
    angle = angle.unsqueeze(-1).unsqueeze(-1)
    eye = torch.eye(3, dtype=axis_unit_vec.dtype).unsqueeze(0)
    z = torch.zeros_like(axis_unit_vec[:, 0])
    ux, uy, uz = axis_unit_vec[:, 0], axis_unit_vec[:, 1], axis_unit_vec[:, 2]
    cross = torch.stack([
        torch.stack([z, -uz, uy], dim=1),
        torch.stack([uz, z, -ux], dim=1),
        torch.stack([-uy, ux, z], dim=1),
    ], dim=1)
    r = torch.cos(angle) * eye
    r = r + torch.sin(angle) * cross
    u = axis_unit_vec.unsqueeze(2)
    r = r + (1 - torch.cos(angle)) * torch.matmul(u, u.transpose(1, 2))
    return r


def dihedral_to_cartesian_one_way(dihedrals: torch.Tensor, cartesian: torch.Tensor) -> torch.Tensor:
    """encodermap_tf1/backmapping.py:198-214 (layers twin misc/backmapping.py:1873-1912)."""
    n = dihedrals.shape[-1]
    dihedrals = -dihedrals
    rotated = cartesian[:, 1:]
    collected = [cartesian[:, 0:1]]
    for i in range(n):
        collected.append(rotated[:, 0:1])
        axis = rotated[:, 1] - rotated[:, 0]
        axis = axis / torch.linalg.norm(axis, dim=1, keepdim=True)
        offset = rotated[:, 1:2]
        rotated = offset + torch.matmul(rotated[:, 1:] - offset, rotation_matrix(axis, dihedrals[:, i]))
    collected.append(rotated)
    return torch.cat(collected, dim=1)


def dihedrals_to_cartesian(dihedrals: torch.Tensor, cartesian: torch.Tensor) -> torch.Tensor:
    """encodermap_tf1/backmapping.py:164-195 (layers twin misc/backmapping.py:259-309)."""
    if cartesian.dim() == 2:
        cartesian = cartesian.unsqueeze(0).expand(dihedrals.shape[0], -1, -1)
    split = int(int(cartesian.shape[1]) / 2)
    cartesian_right = cartesian[:, split - 1:]
    dihedrals_right = dihedrals[:, split - 1:]
    cartesian_left = torch.flip(cartesian[:, : split + 2], dims=[1])      # [:, split+1::-1]
    dihedrals_left = torch.flip(dihedrals[:, : split - 1], dims=[1])      # [:, split-2::-1]
    new_right = dihedral_to_cartesian_one_way(dihedrals_right, cartesian_right)
    new_left = dihedral_to_cartesian_one_way(dihedrals_left, cartesian_left)
    return torch.cat([torch.flip(new_left, dims=[1]), new_right[:, 3:]], dim=1)


def backmap_layer(angles: torch.Tensor, dihedrals: torch.Tensor, inp_distances: torch.Tensor) -> torch.Tensor:
    """BackMapLayer.call (models/layers.py:957-986): batch-mean bond lengths -> chain_in_plane ->
    dihedrals_to_cartesian(dihedrals + pi)."""
    mean_lengths = torch.mean(inp_distances, dim=0, keepdim=True)
    plane = chain_in_plane(mean_lengths, angles)
    return dihedrals_to_cartesian(dihedrals + PI, plane)


def pairwise_distances_layer(xyz: torch.Tensor, start=None, stop=None, step=None) -> torch.Tensor:
    """PairwiseDistances.call (models/layers.py:1252-1267): slice atoms, then pairwise_dist(flat=True)."""
    return pairwise_dist(xyz[:, start:stop:step], flat=True)


def mean_angles_layer(angles: torch.Tensor) -> torch.Tensor:
    """MeanAngles.call (models/layers.py:1153-1161): tile(mean over batch)."""
    return torch.mean(angles, dim=0, keepdim=True).expand(angles.shape[0], -1)


def dihedrals_mdtraj_convention(xyz: torch.Tensor) -> torch.Tensor:
    """Dihedral of consecutive atom quadruplets, the arctan2 formula restated at
    misc/backmapping.py:344-352 (mdtraj convention).  xyz [B, A, 3] -> [B, A-3]."""
    b1 = xyz[:, 1:-2] - xyz[:, :-3]
    b2 = xyz[:, 2:-1] - xyz[:, 1:-2]
    b3 = xyz[:, 3:] - xyz[:, 2:-1]
    c1 = torch.cross(b2, b3, dim=-1)
    c2 = torch.cross(b1, b2, dim=-1)
    p1 = (b1 * c1).sum(-1) * torch.linalg.norm(b2, dim=-1)
    p2 = (c1 * c2).sum(-1)
    return torch.atan2(p1, p2)


# --------------------------------------------------------------------------- #
# a21: ADC functional model -- models/models.py:385-1060 (graph), :2260-2521 (get_loss / train_step)
# --------------------------------------------------------------------------- #
class ADCOracle:
    """gen_functional_model as plain tensors.  ``params`` = [W0, b0, ...] over encoder then decoder Dense layers."""

    def __init__(self, n_atoms: int, n_side: int, p: dict, params: Sequence[np.ndarray], dtype=torch.float64):
        self.p = p
        self.A = n_atoms
        self.use_ang = bool(p.get("use_backbone_angles", False))
        self.use_side = bool(p.get("use_sidechains", False))
        self.periodicity = float(p.get("periodicity", 2 * PI))
        widths = ([n_atoms - 2] if self.use_ang else []) + [n_atoms - 3] + ([n_side] if self.use_side else [])
        self.splits = [2 * w for w in widths]                      # models.py:1980-2002
        net_dim = sum(self.splits)
        enc, dec = layer_plan(p["n_neurons"], p["activation_functions"], net_dim)
        self.enc_acts = [a for _, a in enc]
        self.dec_acts = [a for _, a in dec]
        self.n_enc = len(enc)
        self.params = [torch.tensor(np.asarray(q), dtype=dtype).requires_grad_(True) for q in params]
        self.dtype = dtype

    def encoder(self, inputs: Sequence[torch.Tensor]) -> torch.Tensor:
        x = torch.cat([periodic_input(a, self.periodicity) for a in inputs], dim=1)   # PeriodicInput + Concatenate, :625-630,872-878
        W, b = self.params[0::2], self.params[1::2]
        return dense_stack(x, W[: self.n_enc], b[: self.n_enc], self.enc_acts)

    def decoder(self, z: torch.Tensor):
        W, b = self.params[0::2], self.params[1::2]
        y = dense_stack(z, W[self.n_enc:], b[self.n_enc:], self.dec_acts)
        return [periodic_output(part, self.periodicity) for part in torch.split(y, self.splits, dim=1)]   # :1499-1541

    def kernels(self):
        return self.params[0::2]


def adc_loss(model: ADCOracle, angles, dihedrals, cartesians, distances, side, step: int):
    """ADCFunctionalModel.get_loss (models.py:2260-2460); returns (total, dict of terms)."""
    p = model.p
    per = model.periodicity
    enc_in = ([angles] if model.use_ang else []) + [dihedrals] + ([side] if model.use_side else [])
    latent = model.encoder(enc_in)
    outs = model.decoder(latent)
    if model.use_ang:
        out_angles, out_dihedrals = outs[0], outs[1]
        out_side = outs[2] if model.use_side else None
    else:
        out_angles, out_dihedrals, out_side = mean_angles_layer(angles), outs[0], None
    back = backmap_layer(out_angles, out_dihedrals, distances)
    sl = (p.get("cartesian_pwd_start"), p.get("cartesian_pwd_stop"), p.get("cartesian_pwd_step"))
    inp_pair = pairwise_distances_layer(cartesians, *sl)
    out_pair = pairwise_distances_layer(back, *sl)
    terms = {
        "dihedral": angle_like_cost(dihedrals, out_dihedrals, per, p.get("dihedral_cost_variant", "mean_abs"),
                                    p.get("dihedral_cost_reference", 1), p.get("dihedral_cost_scale", 1)),
        "angle": angle_like_cost(angles, out_angles, per, p.get("angle_cost_variant", "mean_abs"),
                                 p.get("angle_cost_reference", 1), p.get("angle_cost_scale", 0)),
        "cartesian": cartesian_cost(inp_pair, out_pair, p.get("cartesian_cost_variant", "mean_abs"), p.get("cartesian_cost_reference", 1),
                                    soft_start_scale(step, p.get("cartesian_cost_scale_soft_start", (None, None)), p.get("cartesian_cost_scale", 1))),
        "distance": distance_cost(torch.cat(enc_in[:3], dim=1) if len(enc_in) > 1 else dihedrals, latent, per,
                                  p.get("dist_sig_parameters", (4.5, 12, 6, 1, 2, 6)), p.get("distance_cost_scale")),
        "cartesian_distance": (sigmoid_loss(inp_pair, latent, float("inf"), p.get("cartesian_dist_sig_parameters", (4.5, 12, 6, 1, 2, 6)))
                               * p["cartesian_distance_cost_scale"]) if p.get("cartesian_distance_cost_scale", 1) is not None else latent.new_zeros(()),
        "center": center_cost(latent, p.get("center_cost_scale", 1e-4)),
        "reg": regularization_cost(model.kernels(), p.get("l2_reg_constant", 0.001)),
        "side_dihedral": angle_like_cost(side, out_side, per, p.get("side_dihedral_cost_variant", "mean_abs"),
                                         p.get("side_dihedral_cost_reference", 1), p.get("side_dihedral_cost_scale", 0.5))
        if model.use_side else latent.new_zeros(()),
    }
    return sum(terms.values()), terms


def adc_train_steps(model: ADCOracle, data: dict, batch_indices, start_step: int = 0):
    """Oracle ADC training loop with injected weights and batch indices (Adam with clipvalue 1 as autoencoder.py:1664-1666)."""
    opt = KerasAdam(model.params, learning_rate=model.p.get("learning_rate", 1e-3), clipvalue=1.0)
    t = {k: torch.tensor(np.asarray(v), dtype=model.dtype) for k, v in data.items()}
    losses, terms_all = [], []
    for n, idx in enumerate(batch_indices):
        ii = torch.as_tensor(np.asarray(idx), dtype=torch.long)
        total, terms = adc_loss(model, t["central_angles"][ii], t["central_dihedrals"][ii], t["central_cartesians"][ii],
                                t["central_distances"][ii], t["side_dihedrals"][ii] if "side_dihedrals" in t else None, start_step + n)
        grads = torch.autograd.grad(total, model.params)
        opt.apply_gradients(model.params, grads)
        losses.append(float(total.detach()))
        terms_all.append({k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in terms.items()})
    return losses, terms_all


def train_for_references(data: dict, batch_indices, pwd_start=None, pwd_stop=None, pwd_step=None, dtype=torch.float64) -> dict:
    """AngleDihedralCartesianEncoderMap.train_for_references (autoencoder/autoencoder.py:1816-1937): the costs of the trivial
    model that answers every batch with the data-set means, per batch, with the cost closures of a FRESH ADCParameters
    (scales 1, references 1, mean_abs, periodicity 2 pi; :1822-1826):
      dihedral / angle cost   the batch against the column means of the whole data set, repeated to batch size (:1849-1861),
      cartesian cost          pair distances of the mean structure -- chain_in_plane(mean bond lengths, mean angles) turned by
                              the mean dihedrals + pi (:1863-1877), sliced by cartesian_pwd_{start,stop,step} -- against the
                              pair distances of the batch's Cartesians (:1912-1921), plain differences.
    ``batch_indices``: the rows of each batch (the reference iterates its shuffled tf.data set).  Returns the per-batch
    lists; their means are the new ``*_cost_reference`` (:1927-1934)."""
    t = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in data.items()}
    mean_dih = t["central_dihedrals"].mean(0, keepdim=True)
    mean_ang = t["central_angles"].mean(0, keepdim=True)
    mean_len = t["central_distances"].mean(0, keepdim=True)
    chain = chain_in_plane(mean_len, mean_ang)
    gen = dihedrals_to_cartesian(mean_dih + PI, chain)
    pd_mean = pairwise_dist(gen[:, pwd_start:pwd_stop:pwd_step], flat=True)
    out = {"dihedral_cost": [], "angle_cost": [], "cartesian_cost": []}
    for idx in batch_indices:
        ii = torch.as_tensor(np.asarray(idx), dtype=torch.long)
        B = len(ii)
        out["dihedral_cost"].append(float(angle_like_cost(t["central_dihedrals"][ii], mean_dih.expand(B, -1), 2 * PI, "mean_abs", 1.0, 1.0)))
        out["angle_cost"].append(float(angle_like_cost(t["central_angles"][ii], mean_ang.expand(B, -1), 2 * PI, "mean_abs", 1.0, 1.0)))
        pd = pairwise_dist(t["central_cartesians"][ii][:, pwd_start:pwd_stop:pwd_step], flat=True)
        out["cartesian_cost"].append(float(cartesian_cost(pd_mean.expand(B, -1), pd, "mean_abs", 1.0, 1.0)))
    return out


def synthetic_backbone(rng: np.random.Generator, n_frames: int, n_residues: int, n_side: int = 0) -> dict:
    """Synthetic protein-like backbone: dihedrals around two basins, angles N(2.0, 0.05),
    bond lengths N(1.4, 0.02) (arbitrary length unit); central_cartesians are built with THIS file's
    chain_in_plane + dihedrals_to_cartesian so that the inputs are self-consistent."""
    A = 3 * n_residues
    basin = rng.integers(0, 2, (n_frames, 1))
    centers = np.where(basin == 0, -1.0, 2.4)
    dih = centers + rng.normal(0, 0.35, (n_frames, A - 3))
    dih = (dih + PI) % (2 * PI) - PI
    ang = rng.normal(2.0, 0.05, (n_frames, A - 2))
    length = rng.normal(1.4, 0.02, (n_frames, A - 1))
    plane = chain_in_plane(torch.tensor(length), torch.tensor(ang))
    xyz = dihedrals_to_cartesian(torch.tensor(dih) + PI, plane).numpy()
    out = {"central_angles": ang.astype(np.float32), "central_dihedrals": dih.astype(np.float32),
           "central_cartesians": xyz.astype(np.float32), "central_distances": length.astype(np.float32)}
    if n_side:
        out["side_dihedrals"] = rng.uniform(-PI, PI, (n_frames, n_side)).astype(np.float32)
    return out
