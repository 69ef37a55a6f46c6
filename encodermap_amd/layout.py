"""Host-side description of the dense autoencoder and of its flat parameter buffer.

Pure Python/NumPy (no GPU needed): layer stacking rules of the reference's ``SequentialModel``
(encodermap/models/models.py:3121-3160), Keras-compatible initialisers (models.py:3184-3186;
``VarianceScaling()`` / ``RandomNormal(0.1, 0.05)``) and the offsets of every
kernel/bias inside ONE contiguous float32 buffer -- the layout the fused Adam kernel and the single
gradient all-reduce operate on.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

_TRUNC_STD = 0.87962566103423978  # std of a unit normal truncated to [-2, 2]


@dataclass(frozen=True)
class DenseSpec:
    name: str          # Encoder_i / Latent / Decoder_i (models.py:3142-3150)
    n_in: int
    n_out: int
    activation: Optional[str]  # a key of ACTIVATION_CODES, or None (linear)
    w_off: int         # offset of the [n_in, n_out] kernel in the flat buffer
    b_off: int         # offset of the [n_out] bias


@dataclass(frozen=True)
class NetLayout:
    layers: tuple          # all DenseSpec, encoder first
    n_enc: int             # number of encoder layers (index of the first decoder layer)
    net_dim: int           # width the network sees (2*D for periodic input)
    n_params: int          # floats in the flat buffer (multiple of 4 for 16-byte lanes)
    n_real: int            # floats that are actual parameters

    @property
    def encoder(self):
        return self.layers[: self.n_enc]

    @property
    def decoder(self):
        return self.layers[self.n_enc:]

    @property
    def latent_dim(self) -> int:
        return self.layers[self.n_enc - 1].n_out


# Keras activation names (models.py:3129-3131, 3179-3220: any name goes to Dense) -> the codes of emx_mlp_desc::act (include/emx.h)
ACTIVATION_CODES = {"tanh": 1, "relu": 2, "sigmoid": 3, "elu": 4, "softplus": 5, "selu": 6, "softsign": 7, "exponential": 8}
ACTIVATION_ALIASES = {"linear": None, "Tanh": "tanh"}


def build_layout(n_neurons: Sequence[int], activation_functions: Sequence[str], net_dim: int) -> NetLayout:
    """[128,128,2] + ['', 'tanh', 'tanh', ''] -> Encoder_0(tanh) Encoder_1(tanh) Latent(linear) |
    Decoder_0(tanh) Decoder_1(tanh) Decoder_2(linear, net_dim)."""
    if len(n_neurons) != len(activation_functions) - 1:
        raise Exception(
            f"Length of `n_neurons` and `activation_functions` (-1) does not match: {n_neurons}, {activation_functions}")
    acts = [ACTIVATION_ALIASES.get(a, a) if a not in ("", None) else None for a in activation_functions]
    for a in acts:
        if a is not None and a not in ACTIVATION_CODES:
            raise NotImplementedError(f"activation function {a!r} is not available in the HIP engine: its backward kernels keep a layer's OUTPUT, "
                                      f"so they take the Keras activations whose derivative is a function of it -- {sorted(ACTIVATION_CODES)} "
                                      f"or '' (linear); swish / gelu / softmax need the pre-activation")
    units = list(n_neurons) + list(n_neurons[-2::-1]) + [net_dim]
    layer_acts = acts[1:] + acts[-2::-1] + [None]
    n_lat = len(n_neurons)
    bottleneck = units.index(min(units)) + 1
    layers, off, prev = [], 0, net_dim
    for i, (u, a) in enumerate(zip(units, layer_acts)):
        if i < n_lat - 1:
            name = f"Encoder_{i}"
        elif i > n_lat - 1:
            name = f"Decoder_{i - n_lat}"
        else:
            name = "Latent"
        w_off = off
        off += prev * u
        b_off = off
        off += u
        # keep every tensor 16-byte aligned inside the buffer (float4 lanes in the Adam kernel, GEMM alignment)
        off = (off + 3) // 4 * 4
        layers.append(DenseSpec(name, prev, u, a, w_off, b_off))
        prev = u
    n_real = sum(l.n_in * l.n_out + l.n_out for l in layers)
    return NetLayout(tuple(layers), bottleneck, net_dim, off, n_real)


def init_parameters(layout: NetLayout, seed: Optional[int]) -> np.ndarray:
    """Flat float32 buffer initialised like the reference: kernels VarianceScaling() = truncated normal with
    stddev sqrt(1/fan_in)/0.8796 cut at 2 sigma, biases RandomNormal(0.1, 0.05).  Same distributions, own
    stream (TensorFlow's Philox streams are not reproducible outside TensorFlow)."""
    rng = np.random.default_rng(seed)
    flat = np.zeros(layout.n_params, np.float32)
    for l in layout.layers:
        std = np.sqrt(1.0 / l.n_in) / _TRUNC_STD
        n = l.n_in * l.n_out
        w = rng.standard_normal(n)
        bad = np.abs(w) > 2.0
        while bad.any():
            w[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(w) > 2.0
        flat[l.w_off: l.w_off + n] = (w * std).astype(np.float32)
        flat[l.b_off: l.b_off + l.n_out] = rng.normal(0.1, 0.05, l.n_out).astype(np.float32)
    return flat


def l2_coefficients(layout: NetLayout, l2_reg_constant: float) -> np.ndarray:
    """Per-element 2*lambda on kernels, 0 on biases/padding: d/dw of Keras ``l2(lambda) = lambda*sum(w^2)``
    (kernel_regularizer only, models.py:3126,3185)."""
    c = np.zeros(layout.n_params, np.float32)
    for l in layout.layers:
        c[l.w_off: l.w_off + l.n_in * l.n_out] = 2.0 * float(l2_reg_constant)
    return c


def pack_weights(layout: NetLayout, arrays: Sequence[np.ndarray]) -> np.ndarray:
    """[W0, b0, W1, b1, ...] (Keras ``get_weights`` order) -> flat buffer."""
    if len(arrays) != 2 * len(layout.layers):
        raise ValueError(f"expected {2 * len(layout.layers)} arrays (kernel, bias per layer), got {len(arrays)}")
    flat = np.zeros(layout.n_params, np.float32)
    for l, W, b in zip(layout.layers, arrays[0::2], arrays[1::2]):
        W = np.asarray(W, np.float32)
        b = np.asarray(b, np.float32)
        if W.shape != (l.n_in, l.n_out) or b.shape != (l.n_out,):
            raise ValueError(f"{l.name}: expected kernel {(l.n_in, l.n_out)} and bias {(l.n_out,)}, got {W.shape} and {b.shape}")
        flat[l.w_off: l.w_off + W.size] = W.reshape(-1)
        flat[l.b_off: l.b_off + b.size] = b
    return flat


def unpack_weights(layout: NetLayout, flat: np.ndarray) -> list:
    out = []
    for l in layout.layers:
        out.append(np.array(flat[l.w_off: l.w_off + l.n_in * l.n_out]).reshape(l.n_in, l.n_out))
        out.append(np.array(flat[l.b_off: l.b_off + l.n_out]))
    return out


def weights_from_initializer_dict(layout: NetLayout, d: dict) -> np.ndarray:
    """The reference's initializer-dict convention (models.py:113-248): keys ``dense/kernel``, ``dense/bias``,
    ``dense_1/kernel`` ... in layer order."""
    arrays = []
    for i in range(len(layout.layers)):
        stem = "dense" if i == 0 else f"dense_{i}"
        arrays += [d[f"{stem}/kernel"], d[f"{stem}/bias"]]
    return pack_weights(layout, arrays)


def _mix32(h):
    """murmur3 finaliser on uint32 arrays (the round function of feistel_permutation)."""
    h = np.asarray(h, np.uint32).copy()
    h ^= h >> np.uint32(16)
    h *= np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h *= np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return h


def feistel_permutation(n: int, seed: int, epoch: int) -> np.ndarray:
    """Host statement of ``emx_permute`` (include/emx.h), bit for bit: a pseudo-random permutation of 0 .. n-1 chosen by
    (seed, epoch) -- a 6-round Feistel network on the ceil(log2 n) bits of the index, cycle-walked until the value falls
    below n."""
    n = int(n)
    seed, epoch = int(seed) & (2 ** 64 - 1), int(epoch) & (2 ** 64 - 1)
    bits = 2
    while (1 << bits) < n:
        bits += 1
    abits = bits // 2
    bbits = bits - abits
    amask, bmask = np.uint64((1 << abits) - 1), np.uint64((1 << bbits) - 1)
    with np.errstate(over="ignore"):
        u32 = lambda v: np.uint32(v & 0xFFFFFFFF)
        keys = [_mix32(u32(seed) ^ _mix32(u32((seed >> 32) + 0x9E3779B9 * (k + 1))) ^ _mix32(u32((epoch & 0xFFFFFFFF) + 0x7F4A7C15 * (k + 1))) ^
                       _mix32(u32((epoch >> 32) ^ 0x165667B1))) for k in range(6)]
        x = np.arange(n, dtype=np.uint64)
        todo = np.ones(n, bool)
        while todo.any():
            v = x[todo]
            l, r = v >> np.uint64(bbits), v & bmask
            for k in range(0, 6, 2):
                l = (l ^ _mix32((r & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ keys[k]).astype(np.uint64)) & amask
                r = (r ^ _mix32((l & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ keys[k + 1]).astype(np.uint64)) & bmask
            v = (l << np.uint64(bbits)) | r
            x[todo] = v
            todo[todo] = v >= np.uint64(n)
    return x.astype(np.int64)


class BatchIndexStream:
    """Row indices for ``Dataset.shuffle(N, reshuffle_each_iteration=True).repeat().batch(B)``
    (encodermap/autoencoder/autoencoder.py:845-856): a fresh permutation per pass, batches always full and allowed to
    straddle passes.  Host-side statement of the stream the engine draws on the device (``emx_permute`` with the same seed,
    one epoch per pass): the two are identical index for index."""

    def __init__(self, n_rows: int, batch_size: int, seed: Optional[int] = None):
        self.n, self.b = int(n_rows), int(batch_size)
        self.seed = int(np.random.SeedSequence().entropy & (2 ** 63 - 1)) if seed is None else int(seed)
        self.epoch = 0
        self._left = np.empty(0, np.int64)

    def next_block(self, n_batches: int) -> np.ndarray:
        need = n_batches * self.b
        parts = [self._left]
        have = self._left.size
        while have < need:
            p = feistel_permutation(self.n, self.seed, self.epoch)
            self.epoch += 1
            parts.append(p)
            have += p.size
        allidx = np.concatenate(parts)
        self._left = allidx[need:]
        return allidx[:need].reshape(n_batches, self.b)
