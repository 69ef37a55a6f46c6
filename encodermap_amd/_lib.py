"""ctypes binding of libemx.so (C ABI declared in include/emx.h).

The product path has NO fallback: if the shared library is missing or a call is rejected this module
raises.  ``import torch`` must happen before the library is loaded so that libemx resolves the HIP
runtime torch already mapped (both carry the soname ``libamdhip64.so.7``), which makes
``torch.cuda.current_stream().cuda_stream`` a valid ``hipStream_t`` for the kernels here.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMX_LIB_PATH") or os.path.join(_HERE, "libemx.so")   # EMX_LIB_PATH: A/B runs of two builds on one box

EMX_REDUCE_WS = 1024
MEAN_ABS, MEAN_SQUARE, MEAN_NORM = 0, 1, 2
VARIANTS = {"mean_abs": MEAN_ABS, "mean_square": MEAN_SQUARE, "mean_norm": MEAN_NORM}


EMX_EUNSUPPORTED = -2   # include/emx.h: the launch plan does not fit (LDS images, unit count, latent width): try another plan


class EmxError(RuntimeError):
    """A rejected libemx call; ``code`` is the entry point's return value (0 for errors raised by the binding itself)."""

    def __init__(self, msg, code: int = 0):
        super().__init__(msg)
        self.code = int(code)

    @property
    def unsupported(self) -> bool:
        return self.code == EMX_EUNSUPPORTED


_f = C.c_float
_i = C.c_int
_l = C.c_int64
_p = C.c_void_p
_sz = C.c_size_t

# name -> (restype, argtypes); kept in one table so tests can check it against include/emx.h
SIGNATURES = {
    "emx_version": (_i, []),
    "emx_last_error": (C.c_char_p, []),
    "emx_target_arch": (C.c_char_p, []),
    "emx_launch_count": (_l, []),
    "emx_transposed_chain_count": (_l, []),
    "emx_forget_counters": (None, [_p]),
    "emx_sketchmap_ws_bytes": (_sz, [_i, _i]),
    "emx_sketchmap_fwd_bwd": (_i, [_p, _i, _i, _i, _p, _i, _f, _f, C.POINTER(_f), _f, _f, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "emx_sketchmap_tiles": (_i, [_p, _i, _i, _i, _p, _i, _f, _f, C.POINTER(_f), _i, _i, _p, _sz, _p, _p]),
    "emx_sketchmap_split_bytes": (_sz, [_i, _i]),
    "emx_sketchmap_split": (_i, [_p, _i, _i, _i, _p, _p]),
    "emx_sketchmap_tiles_dealt": (_i, [_p, _i, _i, _i, _p, _i, _f, _f, C.POINTER(_f), _p, _sz, _p, _i, _i, _p]),
    "emx_sketchmap_reduce_dealt": (_i, [_i, _i, _p, _i, _f, C.POINTER(_f), _f, _f, _i, _i, _i, _i, _p, _p, _sz, _p]),
    "emx_sketchmap_finish": (_i, [_i, _i, _p, _i, _f, C.POINTER(_f), _f, _f, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "emx_sketchmap_describe": (_i, [_i, _i, _i, _p, _p, _i, _f, C.POINTER(_f), _f, _f, _i, _i, _p, _sz, _p]),
    "emx_pairwise_dist": (_i, [_p, _i, _i, _i, _f, _p, _p]),
    "emx_gather_rows": (_i, [_p, _l, _i, _p, _p, _i, _f, _p, _p, _p, _p]),
    "emx_periodic_in": (_i, [_p, _i, _i, _i, _f, _p, _i, _p]),
    "emx_periodic_out": (_i, [_p, _i, _i, _i, _i, _i, _f, _p, _i, _p]),
    "emx_periodic_cost_fwd_bwd": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _i, _f, _i, _f, _p, _p, _i, _i, _p, _p, _p, _p]),
    "emx_center_cost_fwd_bwd": (_i, [_p, _i, _i, _f, _i, _i, _p, _p, _i, _p]),
    "emx_periodic_cost_batch": (_i, [_p, _i, _i, _i, _i, _p]),
    "emx_adam_step": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i, _p]),
    "emx_adam_step_slices": (_i, [_p, _p, _i, _l, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p, _i, _p]),
    "emx_l2_partials": (_i, [_p, _p, _l, _p, _p]),
    "emx_adc_batch_head": (_i, [_p, _p, _l, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _f, _p, _i, _p, _f, _f, _f, _p, _p, _p]),
    "emx_adam_step_history": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _i, _p, _p, _l, _p, _p]),
    "emx_adam_tick": (_i, [_p, _p, _p, _i, _l, _p, _p, _p, _l, _f, _f, _f, _f, _f, _p, _p, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p]),
    "emx_mlp_wgrad_slices": (_i, [_i]),
    "emx_permute": (_i, [_p, _l, C.c_uint64, C.c_uint64, _p]),
    "emx_finish_step": (_i, [_p, _i, _p, _i, _p, _p, _l, _p, _p, _p, _p]),
    "emx_step_tick": (_i, [_p, _i, _p, _p, _l, _p, _p, _p, _p]),
    "emx_colmean": (_i, [_p, _i, _i, _i, _p, _p]),
    "emx_chain_in_plane_fwd": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "emx_chain_in_plane_bwd": (_i, [_p, _p, _i, _p, _i, _i, _p, _i, _p]),
    "emx_dihedrals_to_cartesian_fwd": (_i, [_p, _i, _f, _p, _i, _i, _p, _p]),
    "emx_dihedrals_to_cartesian_bwd": (_i, [_p, _i, _f, _p, _p, _i, _i, _p, _i, _p, _p]),
    "emx_pdist_flat_fwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p]),
    "emx_pdist_flat_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p]),
    "emx_adc_head_bytes": (_sz, []),
    "emx_adc_head_describe": (_i, [_p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _i, _f, _p, _i, _p, _f, _f, _f, _p, _p, _i, _p]),
    "emx_adc_frame_bwd": (_i, [_p, _p, _i, _i, _f, _p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _p, _i, _p]),
    "emx_adc_frame_fwd": (_i, [_p, _i, _i, _f, _i, _i, _p, _p, _i, _i, _p, _p, _f, _i, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p, _p]),
    "emx_adc_frame_fwd_bwd": (_i, [_p, _i, _i, _f, _i, _i, _p, _p, _i, _i, _p, _p, _f, _i, _p, _p, _p, _i, _i, _i, _p, _p, _i, _p,
                                   _i, _f, _p, _p, _p, _i, _p, _p, _i, _p, _i, _p]),
    "emx_backbone_features": (_i, [_p, _l, _i, _p, _p, _p, _p]),
    "emx_soft_start_scale": (_i, [_p, _f, _f, _f, _p, _p]),
    "emx_allatom_mask_words": (_i, [_i]),
    "emx_allatom_backmap": (_i, [_p, _l, _l, _i, _p, _p, _i, _p, _l, _p, _p]),
    "emx_mlp_mirror_floats": (_sz, [_p]),
    "emx_mlp_mirror": (_i, [_p, _p, _p, _p]),
    "emx_mlp_aop_bytes": (_sz, [_p, _i]),
    "emx_mlp_forward": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _p, _p, _i, _p]),
    "emx_mlp_infer": (_i, [_p, _p, _i, _i, _p, _l, _f, _l, _p, _l, _f, _p]),
    "emx_p2p_handle_bytes": (_i, []),
    "emx_p2p_create": (_i, [_i, _i, _l, _i, C.POINTER(_p), _p]),
    "emx_p2p_connect": (_i, [_p, _p]),
    "emx_p2p_allreduce": (_i, [_p, _p, _l, _f, _p]),
    "emx_p2p_status": (_i, [_p, C.POINTER(_i)]),
    "emx_p2p_destroy": (_i, [_p]),
    "emx_mlp_row_blocks": (_i, [_i]),
    "emx_mlp_forward_cost": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _p, _p, _p, _i, _p]),
    "emx_clash_count": (_i, [_p, _i, _i, _f, _p, _p]),
    "emx_kabsch_rmsd": (_i, [_p, _p, _i, _i, C.POINTER(_f), _i, _p, _p]),
    "emx_metric_tick": (_i, [_p, _p, _i, _p, _p, _l, _p, _p]),
    "emx_mlp_forward_backward": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _p, _p, _i, _p, _p, _p, _i, _p]),
    "emx_mlp_forward_backward_tiles": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _p, _p, _i, _p, _p, _p, _i, _p, _i, _i, _f, _f, C.POINTER(_f), _p, _sz, _p,
                                            _p, _p, _p]),
    "emx_mlp_forward_backward_tiles_join": (_i, [_p, _p, _p, _i, _i, _p, _i, _i, _i, _p, _p, _i, _p, _p, _p, _i, _p, _i, _i, _f, _f, C.POINTER(_f), _p, _sz, _p,
                                                 _p, _p, _i, _p, _p]),
    "emx_mlp_backward": (_i, [_p, _p, _p, _i, _i, _p, _i, _p, _i, _p, _p, _i, _i, _p, _p, _p, _i, _p]),
    "emx_mlp_wgrad_ws_floats": (_sz, [_l, _i]),
    "emx_mlp_wgrad": (_i, [_p, _p, _l, _i, _i, _p, _i, _i, _i, _p, _p, _sz, _p, _p]),
    "emx_mlp_wgrad_adam_local_fits": (_i, [_p, _i, _p]),
    "emx_mlp_wgrad_adam_local_ws_floats": (_sz, [_p, _i]),
    "emx_mlp_wgrad_adam_local": (_i, [_p, _p, _l, _p, _i, _i, _i, _p, _sz, _p, _p, _p, _p, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
}

EMX_MAX_LAYERS = 12
EMX_MAX_SUM_JOBS = 8
EMX_MAX_COST_JOBS = 4


class LatentGrad(C.Structure):
    """Mirror of ``emx_latent_grad``."""

    _fields_ = [("partials", _p), ("Z", _p), ("slot_stride", _l), ("n_slots", _i), ("row_base", _i), ("gscale", _f), ("center_grad", _f)]


class NextBatch(C.Structure):
    """Mirror of ``emx_next_batch``: the rows of batch cursor + 1, gathered beside the optimizer."""

    _fields_ = [("data", _p), ("n_rows", _l), ("D", _i), ("idx", _p), ("n_batches", _i), ("cursor", _p), ("B", _i), ("periodicity", _f),
                ("xb", _p), ("xin", _p), ("split_ws", _p), ("adc_head", _p)]


def make_next_batch(data, idx, cursor, B, periodicity, xb, xin, split_ws=None) -> NextBatch:
    N, D = data.shape
    assert idx.dtype == torch.int64 and idx.is_contiguous() and data.is_contiguous() and idx.numel() % B == 0
    nb = NextBatch(_ptr(data), N, D, _ptr(idx), idx.numel() // B, _ptr(cursor), int(B), float(periodicity), _ptr(xb), _ptr(xin),
                   _ptr(split_ws), None)
    nb.keep = (data, idx, cursor, xb, xin, split_ws)
    return nb


def make_next_adc_head(record, idx, cursor, B, mean_columns) -> NextBatch:
    """The head of the next ADC step beside the optimizer: ``record`` from ``adc_head_describe`` (a uint8 device tensor)."""
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() % B == 0
    nb = NextBatch(None, 0, int(mean_columns), _ptr(idx), idx.numel() // B, _ptr(cursor), int(B), 0.0, None, None, None, _ptr(record))
    nb.keep = (record, idx, cursor)
    return nb


class AutoCost(C.Structure):
    """Mirror of ``emx_auto_cost``: the auto cost folded into the forward launch."""

    _fields_ = [("target", _p), ("ldt", _i), ("n", _i), ("off_sin", _i), ("off_cos", _i), ("periodic_output", _i), ("periodicity", _f),
                ("variant", _i), ("coef", _f), ("B", _i), ("dY", _p), ("ws", _p)]


def make_auto_cost(target, n, off_sin, off_cos, periodic_output, periodicity, variant, coef, B, dY, ws) -> AutoCost:
    c = AutoCost(_ptr(target), target.stride(0), int(n), int(off_sin), int(off_cos), int(bool(periodic_output)), float(periodicity),
                 VARIANTS[variant] if isinstance(variant, str) else int(variant), float(coef), int(B), _ptr(dY), _ptr(ws))
    c.keep = (target, dY, ws)
    return c


def _activation_code(name) -> int:
    from .layout import ACTIVATION_CODES      # (layout.py is the one table of names; include/emx.h: emx_mlp_desc::act)

    return ACTIVATION_CODES[name] if name else 0


class SketchmapPartials(C.Structure):
    """Mirror of ``emx_sketchmap_partials``."""

    _fields_ = [("grad", LatentGrad), ("tile_loss", _p), ("n_tiles", _i), ("loss_scale", _f), ("center_loss_scale", _f)]


class SumJob(C.Structure):
    """Mirror of ``emx_sum_job``."""

    _fields_ = [("src", _p), ("n", _l), ("scale", _f), ("dev_scale", _p), ("slot", _i), ("square", _i)]


class FinishDesc(C.Structure):
    """Mirror of ``emx_finish_desc``: the arguments of emx_finish_step for launches that carry it along."""

    _fields_ = [("jobs", _p), ("n_jobs", _i), ("terms", _p), ("n_terms", _i), ("history", _p), ("term_history", _p), ("capacity", _l),
                ("hstep", _p), ("cursor", _p), ("counter2", _p)]


def _job_array(jobs):
    arr = (SumJob * max(len(jobs), 1))()
    for j, (src, n, scale, dsc, slot, sq) in enumerate(jobs):
        arr[j].src = src if isinstance(src, int) else _ptr(src)
        arr[j].n = int(n)
        arr[j].scale = float(scale)
        arr[j].dev_scale = _ptr(dsc)
        arr[j].slot = int(slot)
        arr[j].square = int(bool(sq))
    return arr


def make_finish_desc(jobs, terms, n_terms, history=None, term_history=None, hstep=None, cursor=None, counter2=None) -> FinishDesc:
    """jobs: list of (src_ptr_or_tensor, n, scale, dev_scale_tensor_or_None, slot, square), as for finish_step."""
    arr = _job_array(jobs)
    d = FinishDesc(C.cast(arr, C.c_void_p), len(jobs), _ptr(terms), int(n_terms), _ptr(history), _ptr(term_history),
                   history.numel() if history is not None else 0, _ptr(hstep), _ptr(cursor), _ptr(counter2))
    d.keep = (arr, jobs, terms, history, term_history, hstep, cursor, counter2)
    return d


class MlpDesc(C.Structure):
    """Mirror of ``emx_mlp_desc`` (include/emx.h)."""

    _fields_ = [("n_layers", _i), ("width", _i * (EMX_MAX_LAYERS + 1)), ("act", _i * EMX_MAX_LAYERS),
                ("w_off", _l * EMX_MAX_LAYERS), ("b_off", _l * EMX_MAX_LAYERS)]


class _LayerWindow:
    """consecutive Dense layers of a network, seen as a network of their own (what MlpPlan reads of a layout)"""

    def __init__(self, net_dim, layers, n_enc=None):
        self.net_dim, self.layers, self.n_enc = int(net_dim), list(layers), n_enc


def _window_bounds(n, n_enc):
    """[begin, end) of the windows of a network of n > EMX_MAX_LAYERS layers: up to EMX_MAX_LAYERS layers each, never across the
    encoder / decoder boundary (the engines launch the two halves separately)"""
    cuts, a = [], 0
    for stop in ([n_enc, n] if n_enc is not None and 0 < n_enc < n else [n]):
        while a < stop:
            b = min(a + EMX_MAX_LAYERS, stop)
            cuts.append((a, b))
            a = b
    return cuts


class MlpPlan:
    """Descriptor + pointer tables for one network (built once per engine).

    ``mirror``: the weight operand store (include/emx.h, emx_mlp_mirror), shared between plans over the same weights.
    ``train_rows``: > 0 allocates the weight-gradient operand store for batches of that many rows (emx_mlp_aop_bytes);
    forward / backward launches then leave their fragments there and ``mlp_wgrad`` consumes them.  0: inference only.
    ``aop``: share another plan's store (a second pointer table over the same network: h[l] / delta[l] = None are rows the
    launches do not write to HBM).

    More than EMX_MAX_LAYERS Dense layers (the reference stacks any number, models/models.py:3133-3160): the plan is a list of
    WINDOWS -- plans of up to EMX_MAX_LAYERS consecutive layers over the same parameter buffer, each with its own descriptor,
    operand store and fragment store (``windows``: (begin, end, plan)).  mlp_forward / mlp_backward / mlp_wgrad walk the windows
    of their layer range (a window's backward launch hands dh of the layer below it to the next through ``dh``); the launches
    that merge passes (mlp_forward_backward*, mlp_wgrad_adam*) answer EMX_EUNSUPPORTED, which is what makes the engines take
    their separate-launch step, and the optimizer launches run without a store, which the windows rebuild afterwards."""

    windows = None

    def __init__(self, layout, params, h, delta, mirror=None, train_rows: int = 0, aop=None):
        n = len(layout.layers)
        self.params = params
        self.n_params = params.numel()
        self.keep = (h, delta)
        if n > EMX_MAX_LAYERS:
            bounds = _window_bounds(n, getattr(layout, "n_enc", None))
            self.windows = []
            for i, (a, b) in enumerate(bounds):
                view = _LayerWindow(layout.net_dim if a == 0 else layout.layers[a - 1].n_out, layout.layers[a:b])
                self.windows.append((a, b, MlpPlan(view, params, list(h[a:b]), list(delta[a:b]), mirror=None if mirror is None else mirror[i],
                                                   train_rows=train_rows)))
            self.mirror = tuple(w.mirror for _, _, w in self.windows)
            self.desc = None
            self.aop_rows = int(train_rows)
            self.aop = tuple(w.aop for _, _, w in self.windows) if train_rows > 0 else None
            # dh of the layer below a window (rows of h[begin - 1]): written by the window's backward launch, read by the next one's
            self.dh = {a: torch.empty_like(h[a - 1]) for a, _, _ in self.windows[1:] if h[a - 1] is not None and delta[a] is not None}
            return
        d = MlpDesc()
        d.n_layers = n
        d.width[0] = layout.net_dim
        for i, lyr in enumerate(layout.layers):
            d.width[i + 1] = lyr.n_out
            d.act[i] = _activation_code(lyr.activation)
            d.w_off[i] = lyr.w_off
            d.b_off[i] = lyr.b_off
        self.desc = d
        PtrArr = _p * EMX_MAX_LAYERS
        self.h = PtrArr(*[(t.data_ptr() if t is not None else None) for t in h] + [None] * (EMX_MAX_LAYERS - n))
        self.delta = PtrArr(*[(t.data_ptr() if t is not None else None) for t in delta] + [None] * (EMX_MAX_LAYERS - n))
        if mirror is None:
            mirror = torch.zeros(load().emx_mlp_mirror_floats(C.byref(d)), dtype=torch.float32, device=params.device)
            self.mirror = mirror
            self.refresh_mirror()
        self.mirror = mirror
        self.aop_rows = int(train_rows)
        self.aop = aop
        if self.aop_rows > 0 and aop is None:
            self.aop = torch.zeros(load().emx_mlp_aop_bytes(C.byref(d), self.aop_rows), dtype=torch.uint8, device=params.device)

    def overlapping(self, l0, l1):
        """(window plan, window begin, first layer, end layer) of the windows that hold layers of [l0, l1), in ascending order"""
        return [(w, a, max(a, l0), min(b, l1)) for a, b, w in self.windows if a < l1 and l0 < b]

    def refresh_mirror(self):
        """Rebuild the operand store after `params` was written by anything other than emx_adam_step*."""
        if self.windows is not None:
            for _, _, w in self.windows:
                w.refresh_mirror()
            return
        _check(load().emx_mlp_mirror(C.byref(self.desc), _ptr(self.params), _ptr(self.mirror), _stream()), "emx_mlp_mirror")

    def aop_args(self, train: bool):
        """(aop pointer, aop_rows) for a launch; (NULL, 0) for inference launches or plans without a store."""
        if not train or self.aop is None:
            return None, 0
        return self.aop.data_ptr(), self.aop_rows


_lib = None


def load() -> C.CDLL:
    """Load libemx.so (once).  Raises EmxError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise EmxError(
            f"{LIB_PATH} not found: the HIP kernel library has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
        fn.restype = res
        fn.argtypes = args
    if lib.emx_version() != 1:
        raise EmxError(f"libemx.so version {lib.emx_version()} does not match the Python binding (1)")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise EmxError(f"{what} failed ({rc}): {load().emx_last_error().decode()}", rc)


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise EmxError("encodermap_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False. There is no CPU fallback.")


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "device tensor expected"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise EmxError(f"{name}: contiguous float32 device tensor expected, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


# ------------------------------------------------------------------------------------------------
# thin, shape-checked wrappers (all asynchronous on torch's current stream)
# ------------------------------------------------------------------------------------------------
def sketchmap_ws_bytes(B: int, L: int) -> int:
    return int(load().emx_sketchmap_ws_bytes(B, L))


def sketchmap_fwd_bwd(X, Z, periodicity, sig, scale, center_scale, row_begin, row_end, dist_loss, center_loss, dZ, ws, data_span=None):
    """``data_span``: max - min over the data the rows come from, if known (periodic data spanning less than 1.5 periods run
    the fixed-point tiles; None: the general float form)."""
    lib = load()
    _f32c(Z, "Z"), _f32c(dZ, "dZ")
    assert X.dtype == torch.float32 and X.stride(1) == 1
    B, D = X.shape
    L = Z.shape[1]
    sig_arr = (_f * 6)(*[float(s) for s in sig])
    rc = lib.emx_sketchmap_fwd_bwd(
        _ptr(X), B, D, X.stride(0), _ptr(Z), L, float(periodicity), float(data_span or 0.0), sig_arr, float(scale),
        float(center_scale if center_loss is not None else 0.0), int(row_begin), int(row_end),
        _ptr(dist_loss), _ptr(center_loss), _ptr(dZ), _ptr(ws), ws.numel() * ws.element_size(), _stream())
    _check(rc, "emx_sketchmap_fwd_bwd")


def sketchmap_tiles(X, Z, periodicity, sig, row_begin, row_end, ws, split_ws=None, data_span=None):
    """``split_ws`` (from sketchmap_split / gather_rows / the optimizer's next-batch gather): Gram products on the f16 matrix cores."""
    B, D = X.shape
    sig_arr = (_f * 6)(*[float(s) for s in sig])
    _check(load().emx_sketchmap_tiles(_ptr(X), B, D, X.stride(0), _ptr(Z), Z.shape[1], float(periodicity), float(data_span or 0.0), sig_arr, int(row_begin),
                                      int(row_end), _ptr(ws), ws.numel() * ws.element_size(), _ptr(split_ws), _stream()),
           "emx_sketchmap_tiles")


def sketchmap_tiles_dealt(X, Z, periodicity, sig, ws, split_ws, rank, world, data_span=None):
    """The symmetric tile pass over the whole batch, this rank's share of the upper-triangle tiles (tile t -> rank t % world)."""
    B, D = X.shape
    sig_arr = (_f * 6)(*[float(s) for s in sig])
    _check(load().emx_sketchmap_tiles_dealt(_ptr(X), B, D, X.stride(0), _ptr(Z), Z.shape[1], float(periodicity), float(data_span or 0.0), sig_arr, _ptr(ws),
                                            ws.numel() * ws.element_size(), _ptr(split_ws), int(rank), int(world), _stream()),
           "emx_sketchmap_tiles_dealt")


def sketchmap_reduce_dealt(B, D, Z, periodicity, sig, scale, center_scale, own_begin, own_end, rank, world, dZ, ws):
    """d cost / d z of ALL rows from this rank's tiles (+ the center gradient on its own rows): summed over ranks, the full gradient."""
    sig_arr = (_f * 6)(*[float(s) for s in sig])
    _check(load().emx_sketchmap_reduce_dealt(int(B), int(D), _ptr(Z), Z.shape[1], float(periodicity), sig_arr, float(scale), float(center_scale),
                                             int(own_begin), int(own_end), int(rank), int(world), _ptr(dZ), _ptr(ws),
                                             ws.numel() * ws.element_size(), _stream()), "emx_sketchmap_reduce_dealt")


def sketchmap_split_bytes(B, D):
    return int(load().emx_sketchmap_split_bytes(int(B), int(D)))


def sketchmap_split(X, split_ws):
    B, D = X.shape
    assert split_ws.numel() * split_ws.element_size() >= sketchmap_split_bytes(B, D)
    _check(load().emx_sketchmap_split(_ptr(X), B, D, X.stride(0), _ptr(split_ws), _stream()), "emx_sketchmap_split")


def sketchmap_finish(B, D, Z, sig, scale, center_scale, row_begin, row_end, dist_loss, center_loss, dZ, ws, periodicity=float("inf")):
    """``periodicity``: the value given to ``sketchmap_tiles`` (it selects the tile size, hence the workspace layout)."""
    sig_arr = (_f * 6)(*[float(s) for s in sig])
    _check(load().emx_sketchmap_finish(B, D, _ptr(Z), Z.shape[1], float(periodicity), sig_arr, float(scale),
                                       float(center_scale if center_loss is not None else 0.0), int(row_begin), int(row_end),
                                       _ptr(dist_loss), _ptr(center_loss), _ptr(dZ), _ptr(ws), ws.numel() * ws.element_size(),
                                       _stream()), "emx_sketchmap_finish")


def pairwise_dist(X, periodicity, out):
    B, D = X.shape
    _check(load().emx_pairwise_dist(_ptr(_f32c(X, "X")), B, D, X.stride(0), float(periodicity), _ptr(out), _stream()), "emx_pairwise_dist")


def permute(out, seed: int, epoch: int):
    """out (int64 device tensor of n elements) <- a pseudo-random permutation of 0 .. n-1 chosen by (seed, epoch)."""
    assert out.dtype == torch.int64 and out.is_contiguous()
    _check(load().emx_permute(_ptr(out), out.numel(), int(seed) & (2 ** 64 - 1), int(epoch) & (2 ** 64 - 1), _stream()), "emx_permute")


def gather_rows(data, idx, periodicity, xb, xin, cursor=None, B=None, split_ws=None):
    N, D = data.shape
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    B = idx.numel() if B is None else B
    _check(load().emx_gather_rows(_ptr(_f32c(data, "data")), N, D, _ptr(idx), _ptr(cursor), B, float(periodicity), _ptr(xb), _ptr(xin),
                                  _ptr(split_ws), _stream()), "emx_gather_rows")


def periodic_in(x, periodicity, out):
    B, D = x.shape
    _check(load().emx_periodic_in(_ptr(x), B, D, x.stride(0), float(periodicity), _ptr(out), out.stride(0), _stream()), "emx_periodic_in")


def periodic_out(y, n, off_sin, off_cos, periodicity, out):
    B = y.shape[0]
    _check(load().emx_periodic_out(_ptr(y), B, n, y.stride(0), off_sin, off_cos, float(periodicity), _ptr(out), out.stride(0), _stream()),
           "emx_periodic_out")


def periodic_cost_fwd_bwd(y, off_sin, off_cos, periodic_output, target, n, periodicity, variant, coef, row_begin, row_end,
                          loss, dY, ws, B=None, coef_scale=None, dpred_add=None):
    B = y.shape[0] if B is None else B
    assert dY is None or dY.stride(0) == y.stride(0)
    assert dpred_add is None or (dpred_add.is_contiguous() and dpred_add.shape[1] == n)
    _check(load().emx_periodic_cost_fwd_bwd(
        _ptr(y), y.stride(0), off_sin, off_cos, int(bool(periodic_output)), _ptr(target), target.stride(0), B, n,
        float(periodicity), VARIANTS[variant] if isinstance(variant, str) else int(variant), float(coef), _ptr(coef_scale),
        _ptr(dpred_add), int(row_begin), int(row_end), _ptr(loss), _ptr(dY), _ptr(ws), _stream()), "emx_periodic_cost_fwd_bwd")


class CostJob(C.Structure):
    """Mirror of ``emx_cost_job`` (include/emx.h)."""

    _fields_ = [("y", _p), ("ldy", _i), ("off_sin", _i), ("off_cos", _i), ("periodic_output", _i), ("target", _p), ("ldt", _i), ("n", _i),
                ("periodicity", _f), ("variant", _i), ("coef", _f), ("coef_scale", _p), ("dpred_add", _p), ("dY", _p), ("ws", _p)]


def periodic_cost_batch(jobs, B, row_begin, row_end):
    """Several row-mean costs in one launch.  jobs: dicts with the keyword names of ``periodic_cost_fwd_bwd`` (y, off_sin, off_cos,
    periodic_output, target, n, periodicity, variant, coef, dY, ws, coef_scale, dpred_add); partials go to each job's ws."""
    arr = _cost_jobs(jobs)
    _check(load().emx_periodic_cost_batch(C.cast(arr, C.c_void_p), len(jobs), int(B), int(row_begin), int(row_end), _stream()),
           "emx_periodic_cost_batch")


def _void_p(arr):
    return C.cast(arr, C.c_void_p)


def _cost_jobs(jobs):
    """the ``emx_cost_job`` array of a list of job dicts (see ``periodic_cost_batch``)"""
    arr = (CostJob * len(jobs))()
    for a, j in zip(arr, jobs):
        y, dY, add = j["y"], j.get("dY"), j.get("dpred_add")
        assert dY is None or dY.stride(0) == y.stride(0)
        assert add is None or (add.is_contiguous() and add.shape[1] == j["n"])
        v = j["variant"]
        a.y, a.ldy, a.off_sin, a.off_cos, a.periodic_output = _ptr(y), y.stride(0), int(j["off_sin"]), int(j["off_cos"]), int(bool(j["periodic_output"]))
        a.target, a.ldt, a.n = _ptr(j["target"]), j["target"].stride(0), int(j["n"])
        a.periodicity, a.variant, a.coef = float(j["periodicity"]), VARIANTS[v] if isinstance(v, str) else int(v), float(j["coef"])
        a.coef_scale, a.dpred_add, a.dY, a.ws = _ptr(j.get("coef_scale")), _ptr(add), _ptr(dY), _ptr(j["ws"])
    return arr


def center_cost_fwd_bwd(Z, scale, row_begin, row_end, loss, dZ, accumulate):
    B, L = Z.shape
    _check(load().emx_center_cost_fwd_bwd(_ptr(_f32c(Z, "Z")), B, L, float(scale), row_begin, row_end, _ptr(loss), _ptr(dZ),
                                          int(bool(accumulate)), _stream()), "emx_center_cost_fwd_bwd")


def _mirror_args(plan):
    """(descriptor, operand store) of the network whose kernels live in the parameter buffer, or (NULL, NULL)."""
    if plan is None or plan.windows is not None:    # (windows: the optimizer runs without a store, _after_optimizer rebuilds them)
        return None, None
    return C.byref(plan.desc), _ptr(plan.mirror)


def _after_optimizer(plan):
    """a plan of windows: its operand stores are rebuilt from the new weights (one launch per window)"""
    if plan is not None and plan.windows is not None:
        plan.refresh_mirror()


def _no_merged_launch(plan, what):
    if plan.windows is not None:
        raise EmxError(f"{what}: more than {EMX_MAX_LAYERS} Dense layers run as windows of separate launches", EMX_EUNSUPPORTED)


def l2_partials(w, l2coef, ws):
    _check(load().emx_l2_partials(_ptr(w), _ptr(l2coef), w.numel(), _ptr(ws), _stream()), "emx_l2_partials")


def adam_step(w, g, m, v, l2coef, lr, beta1, beta2, eps, clip, step, reg_loss, ws, plan=None, next_batch=None, bookkeeping_done=False):
    md, mp = _mirror_args(plan)
    nb = C.byref(next_batch) if next_batch is not None else None
    _check(load().emx_adam_step(_ptr(w), _ptr(g), _ptr(m), _ptr(v), _ptr(l2coef), w.numel(), float(lr), float(beta1), float(beta2),
                                float(eps), float(clip if clip is not None else 0.0), _ptr(step), _ptr(reg_loss), _ptr(ws), md, mp,
                                nb, int(bool(bookkeeping_done)), _stream()), "emx_adam_step")
    _after_optimizer(plan)


def adam_step_history(w, g, m, v, l2coef, lr, beta1, beta2, eps, clip, step, ws, terms, n_terms, history, term_history, hstep,
                      plan=None, next_batch=None):
    """The optimizer launch of a data-parallel step: Adam on the all-reduced gradient (bookkeeping already done in front of
    the collective) + the history row of the all-reduced cost terms (emx_adam_step_history)."""
    md, mp = _mirror_args(plan)
    nb = C.byref(next_batch) if next_batch is not None else None
    cap = history.numel() if history is not None else (term_history.shape[0] if term_history is not None else 0)
    _check(load().emx_adam_step_history(_ptr(w), _ptr(g), _ptr(m), _ptr(v), _ptr(l2coef), w.numel(), float(lr), float(beta1),
                                        float(beta2), float(eps), float(clip if clip is not None else 0.0), _ptr(step), _ptr(ws), md, mp,
                                        nb, _ptr(terms), int(n_terms), _ptr(history), _ptr(term_history), cap, _ptr(hstep), _stream()),
           "emx_adam_step_history")
    _after_optimizer(plan)


def adam_step_slices(w, g_partials, n_slices, m, v, l2coef, lr, beta1, beta2, eps, clip, step, ws, plan=None, next_batch=None,
                     bookkeeping_done=False):
    n = w.numel()
    md, mp = _mirror_args(plan)
    nb = C.byref(next_batch) if next_batch is not None else None
    _check(load().emx_adam_step_slices(_ptr(w), _ptr(g_partials), int(n_slices), n, _ptr(m), _ptr(v), _ptr(l2coef), n, float(lr),
                                       float(beta1), float(beta2), float(eps), float(clip if clip is not None else 0.0), _ptr(step),
                                       _ptr(ws), md, mp, nb, int(bool(bookkeeping_done)), _stream()), "emx_adam_step_slices")
    _after_optimizer(plan)


def adam_tick(w, g, g_partials, n_slices, m, v, l2coef, lr, beta1, beta2, eps, clip, opt_step, terms, n_terms, reg_slot, history,
              term_history, hstep, cursor, ws, plan=None):
    n = w.numel()
    md, mp = _mirror_args(plan)
    _check(load().emx_adam_tick(_ptr(w), _ptr(g), _ptr(g_partials), int(n_slices), n, _ptr(m), _ptr(v), _ptr(l2coef), n, float(lr),
                                float(beta1), float(beta2), float(eps), float(clip if clip is not None else 0.0), _ptr(opt_step),
                                _ptr(terms), int(n_terms), int(reg_slot), _ptr(history), _ptr(term_history), history.numel(),
                                _ptr(hstep), _ptr(cursor), _ptr(ws), md, mp, _stream()), "emx_adam_tick")
    _after_optimizer(plan)


def mlp_wgrad_slices(n_rows):
    return int(load().emx_mlp_wgrad_slices(int(n_rows)))


def step_tick(terms, n_terms, history, term_history, step, cursor=None, counter2=None):
    cap = history.numel() if history is not None else (term_history.shape[0] if term_history is not None else 0)
    _check(load().emx_step_tick(_ptr(terms), n_terms, _ptr(history), _ptr(term_history), cap, _ptr(step), _ptr(cursor), _ptr(counter2),
                                _stream()), "emx_step_tick")


def mlp_forward(plan: MlpPlan, l0, l1, x, row_begin, row_end, cost: "AutoCost" = None, train: bool = True):
    """Layers [l0, l1) on rows [row_begin, row_end); ``cost`` evaluates the auto cost on the last layer's rows in the same launch.
    ``train``: leave the weight-gradient fragments in the plan's store (if it has one; the row range must then start at a
    multiple of 16)."""
    if plan.windows is not None:
        for w, a, b0, b1 in plan.overlapping(l0, l1):
            mlp_forward(w, b0 - a, b1 - a, x if b0 == l0 else plan.keep[0][b0 - 1], row_begin, row_end, cost if b1 == l1 else None, train)
        return
    ap, ar = plan.aop_args(train)
    _check(load().emx_mlp_forward_cost(C.byref(plan.desc), _ptr(plan.params), _ptr(plan.mirror), l0, l1, _ptr(x), x.stride(0), int(row_begin),
                                       int(row_end), plan.h, C.byref(cost) if cost is not None else None, ap, ar, _stream()),
           "emx_mlp_forward")


def mlp_infer(plan: MlpPlan, l0, l1, x, out, periodicity_in=0.0, periodicity_out=0.0):
    """Inference: layers [l0, l1) on every row of ``x`` into ``out`` as one streaming launch (include/emx.h, emx_mlp_infer).
    ``periodicity_in``: ``x`` holds raw angles, the sin / cos input is formed on chip; ``periodicity_out``: ``out`` receives the angles
    atan2(sin half, cos half).  Raises EmxError with ``.unsupported`` for networks the streaming pass does not take."""
    if plan.windows is not None:
        raise EmxError("emx_mlp_infer: more than 12 Dense layers run as windows of separate launches", EMX_EUNSUPPORTED)
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.stride(1) == 1 and out.stride(1) == 1 and x.shape[0] == out.shape[0]
    pin = float(periodicity_in) if (periodicity_in and periodicity_in != float("inf")) else 0.0
    pout = float(periodicity_out) if (periodicity_out and periodicity_out != float("inf")) else 0.0
    _check(load().emx_mlp_infer(C.byref(plan.desc), _ptr(plan.params), int(l0), int(l1), _ptr(x), x.stride(0), pin, x.shape[0], _ptr(out),
                                out.stride(0), pout, _stream()), "emx_mlp_infer")


class P2PAllReduce:
    """The one-shot all-reduce of a small buffer over peer-mapped memory (include/emx.h: emx_p2p_*; csrc/p2p.hip) between the ranks of
    one node.  ``exchange(handle: bytes) -> list of every rank's handle in rank order`` is the caller's transport (the engine:
    ``torch.distributed.all_gather_object``).  ``__call__(flat, scale)`` reduces in place on torch's current stream."""

    def __init__(self, world: int, rank: int, n_floats: int, exchange, chunks: int = 64):
        lib = load()
        self.world, self.rank, self.n = int(world), int(rank), int(n_floats)
        hb = lib.emx_p2p_handle_bytes()
        handle = C.create_string_buffer(hb)
        ctx = _p()
        _check(lib.emx_p2p_create(self.world, self.rank, self.n, int(chunks), C.byref(ctx), handle), "emx_p2p_create")
        self._ctx = ctx
        handles = exchange(bytes(handle.raw))
        if len(handles) != self.world or any(len(h) != hb for h in handles):
            raise EmxError(f"P2PAllReduce: {len(handles)} handles for {self.world} ranks")
        _check(lib.emx_p2p_connect(self._ctx, C.c_char_p(b"".join(handles))), "emx_p2p_connect")

    def __call__(self, flat, scale: float = 1.0) -> None:
        assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() <= self.n
        _check(load().emx_p2p_allreduce(self._ctx, _ptr(flat), flat.numel(), float(scale), _stream()), "emx_p2p_allreduce")

    def gave_up(self) -> bool:
        """(synchronises) a wait for a peer ended without the peer: every result since is invalid"""
        v = _i(0)
        _check(load().emx_p2p_status(self._ctx, C.byref(v)), "emx_p2p_status")
        return bool(v.value)

    def close(self) -> None:
        if getattr(self, "_ctx", None):
            load().emx_p2p_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mlp_forward_backward(plan: MlpPlan, l0, l1, x, row_begin, row_end, cost: "AutoCost", bwd_begin, dbelow=None, train: bool = True):
    """Forward [l0, l1), auto cost, backward-data through [bwd_begin, l1) in one launch (delta[l1 - 1] = d cost / d output)."""
    _no_merged_launch(plan, "emx_mlp_forward_backward")
    ap, ar = plan.aop_args(train)
    _check(load().emx_mlp_forward_backward(C.byref(plan.desc), _ptr(plan.params), _ptr(plan.mirror), l0, l1, _ptr(x), x.stride(0),
                                           int(row_begin), int(row_end), plan.h, C.byref(cost), int(bwd_begin), plan.delta, _ptr(dbelow),
                                           ap, ar, _stream()), "emx_mlp_forward_backward")


def mlp_forward_backward_tiles(plan: MlpPlan, l0, l1, x, row_begin, row_end, cost: "AutoCost", bwd_begin, dbelow, X, sig, sk_ws, split_ws, zpub, sync,
                               periodicity=float("inf"), data_span=None, train: bool = True):
    """mlp_forward_backward with the all-pairs tiles of the sketch-map cost as extra workgroups of the same launch (emx.h):
    ``X`` the batch rows [B, D], ``split_ws`` their split planes (None for periodic rows), ``sk_ws`` the tile workspace (as sketchmap_tiles), ``zpub`` [B, 2]
    float32 scratch, ``sync`` 4 x int32 zeroed once."""
    _no_merged_launch(plan, "emx_mlp_forward_backward_tiles")
    ap, ar = plan.aop_args(train)
    sig_arr = (_f * 6)(*[float(v) for v in sig])
    assert sync.dtype == torch.int32 and sync.numel() >= 4 and zpub.dtype == torch.float32
    _check(load().emx_mlp_forward_backward_tiles(C.byref(plan.desc), _ptr(plan.params), _ptr(plan.mirror), l0, l1, _ptr(x), x.stride(0),
                                                 int(row_begin), int(row_end), plan.h, C.byref(cost), int(bwd_begin), plan.delta, _ptr(dbelow),
                                                 ap, ar, _ptr(X), X.shape[1], X.stride(0), float(periodicity), float(data_span or 0.0), sig_arr, _ptr(sk_ws),
                                                 sk_ws.numel() * sk_ws.element_size(),
                                                 _ptr(split_ws), _ptr(zpub), _ptr(sync), _stream()), "emx_mlp_forward_backward_tiles")


def mlp_forward_backward_tiles_join(plan: MlpPlan, l0, l1, x, row_begin, row_end, cost: "AutoCost", bwd_begin, X, sig, sk_ws, split_ws, zpub, sync,
                                    latent_layer, latent, periodicity=float("inf"), data_span=None, train: bool = True):
    """mlp_forward_backward_tiles with the backward units below the latent layer in the same launch (emx.h): ``latent`` = the
    tiles' latent-gradient description (``sketchmap_describe(...).grad``), ``sync`` 8 x int32 zeroed once."""
    _no_merged_launch(plan, "emx_mlp_forward_backward_tiles_join")
    ap, ar = plan.aop_args(train)
    sig_arr = (_f * 6)(*[float(v) for v in sig])
    assert sync.dtype == torch.int32 and sync.numel() >= 8 and zpub.dtype == torch.float32
    _check(load().emx_mlp_forward_backward_tiles_join(C.byref(plan.desc), _ptr(plan.params), _ptr(plan.mirror), l0, l1, _ptr(x), x.stride(0),
                                                      int(row_begin), int(row_end), plan.h, C.byref(cost), int(bwd_begin), plan.delta, None,
                                                      ap, ar, _ptr(X), X.shape[1], X.stride(0), float(periodicity), float(data_span or 0.0), sig_arr,
                                                      _ptr(sk_ws), sk_ws.numel() * sk_ws.element_size(), _ptr(split_ws), _ptr(zpub), _ptr(sync),
                                                      int(latent_layer), C.byref(latent), _stream()), "emx_mlp_forward_backward_tiles_join")


def clash_count(xyz, clash_distance, counts):
    """xyz [B, A, 3] float32 (contiguous), counts [B] int64."""
    B, A = xyz.shape[0], xyz.shape[1]
    assert xyz.is_contiguous() and counts.dtype == torch.int64
    _check(load().emx_clash_count(_ptr(xyz), B, A, float(clash_distance), _ptr(counts), _stream()), "emx_clash_count")


def kabsch_rmsd(a, b, weights3, translate, out):
    """a, b [B, A, 3] float32 (contiguous), out [B] float32."""
    B, A = a.shape[0], a.shape[1]
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    w = (_f * 3)(*[float(x) for x in weights3])
    _check(load().emx_kabsch_rmsd(_ptr(a), _ptr(b), B, A, w, int(bool(translate)), _ptr(out), _stream()), "emx_kabsch_rmsd")


def metric_tick(counts, rmsd, B, clash_hist, rmsd_hist, hstep):
    cap = (clash_hist if clash_hist is not None else rmsd_hist).numel()
    _check(load().emx_metric_tick(_ptr(counts), _ptr(rmsd), int(B), _ptr(clash_hist), _ptr(rmsd_hist), cap, _ptr(hstep), _stream()),
           "emx_metric_tick")


def mlp_row_blocks(n_rows):
    return int(load().emx_mlp_row_blocks(int(n_rows)))


def mlp_backward(plan: MlpPlan, l0, l1, dtop, dadd, layer_add, row_begin, row_end, dbelow=None, latent=None, train: bool = True):
    """delta[l] for l = l1 - 1 .. l0 from ``dtop`` = d loss / d h[l1 - 1]; ``dadd`` / ``latent`` join at ``layer_add``; ``dbelow``
    receives dh of the layer below the range (emx.h)."""
    if plan.windows is not None:
        for w, a, b0, b1 in reversed(plan.overlapping(l0, l1)):
            joins = (dadd is not None or latent is not None) and b0 <= layer_add < b1
            mlp_backward(w, b0 - a, b1 - a, dtop if b1 == l1 else plan.dh[b1], dadd if joins else None, layer_add - a if joins else 0,
                         row_begin, row_end, dbelow if b0 == l0 else plan.dh[b0], latent if joins else None, train)
        return
    ap, ar = plan.aop_args(train)
    _check(load().emx_mlp_backward(C.byref(plan.desc), _ptr(plan.params), _ptr(plan.mirror), l0, l1, _ptr(dtop), dtop.stride(0), _ptr(dadd),
                                   int(layer_add), C.byref(latent) if latent is not None else None, _ptr(dbelow), int(row_begin),
                                   int(row_end), plan.h, plan.delta, ap, ar, _stream()), "emx_mlp_backward")


def sketchmap_describe(X, Z, periodicity, sig, scale, center_scale, row_begin, row_end, ws) -> SketchmapPartials:
    B, D = X.shape
    out = SketchmapPartials()
    sig_arr = (_f * 6)(*[float(s) for s in sig])
    _check(load().emx_sketchmap_describe(B, D, X.stride(0), _ptr(X), _ptr(Z), Z.shape[1], float(periodicity), sig_arr, float(scale),
                                         float(center_scale), int(row_begin), int(row_end), _ptr(ws), ws.numel() * ws.element_size(),
                                         C.byref(out)), "emx_sketchmap_describe")
    return out


def finish_step(jobs, terms, n_terms, history=None, term_history=None, hstep=None, cursor=None, counter2=None):
    """jobs: list of (src_ptr_or_tensor, n, scale, dev_scale_tensor_or_None, slot, square)."""
    arr = _job_array(jobs)
    cap = history.numel() if history is not None else 0
    _check(load().emx_finish_step(arr, len(jobs), _ptr(terms), int(n_terms), _ptr(history), _ptr(term_history), cap, _ptr(hstep),
                                  _ptr(cursor), _ptr(counter2), _stream()), "emx_finish_step")


def mlp_wgrad_ws_floats(n_params, n_rows):
    return int(load().emx_mlp_wgrad_ws_floats(int(n_params), int(n_rows)))


def mlp_wgrad(plan: MlpPlan, l0, l1, row_begin, row_end, grad, ws, finish: "FinishDesc" = None):
    """Weight gradients of layers [l0, l1) over rows [row_begin, row_end) from the fragments the forward and backward launches
    left in the plan's store.  grad=None leaves the per-slice partials in ws (consumed by adam_step_slices); ``finish``: the
    step's scalar epilogue rides along as one extra workgroup."""
    if plan.aop is None:
        raise EmxError("mlp_wgrad: the plan has no weight-gradient operand store (MlpPlan(..., train_rows=B))")
    if plan.windows is not None:       # (the slices' partials and the gradient are addressed by the layers' offsets in the ONE parameter buffer)
        for w, a, b0, b1 in plan.overlapping(l0, l1):
            mlp_wgrad(w, b0 - a, b1 - a, row_begin, row_end, grad, ws, finish if b1 == l1 else None)
        return
    _check(load().emx_mlp_wgrad(C.byref(plan.desc), _ptr(plan.params), plan.n_params, l0, l1, plan.aop.data_ptr(), plan.aop_rows,
                                int(row_begin), int(row_end), _ptr(grad), _ptr(ws), ws.numel(),
                                C.byref(finish) if finish is not None else None, _stream()), "emx_mlp_wgrad")


def mlp_wgrad_adam_local_fits(plan: MlpPlan, n_rows, next_batch=None) -> bool:
    """whether emx_mlp_wgrad_adam_local is made for this network / batch (small batches, small networks)"""
    if plan.windows is not None:
        return False
    nb = C.byref(next_batch) if next_batch is not None else None
    return bool(load().emx_mlp_wgrad_adam_local_fits(C.byref(plan.desc), int(n_rows), nb))


def mlp_wgrad_adam_local_ws_floats(plan: MlpPlan, n_rows) -> int:
    if plan.windows is not None:
        return 0
    return int(load().emx_mlp_wgrad_adam_local_ws_floats(C.byref(plan.desc), int(n_rows)))


def mlp_wgrad_adam_local(plan: MlpPlan, row_begin, row_end, ws, finish: "FinishDesc", m, v, l2coef, lr, beta1, beta2, eps, clip, reg_ws, sync,
                         next_batch=None):
    """Weight gradients of all layers without slice partials (a workgroup per 32 x 32 block of a kernel, over all rows), the step's
    scalar epilogue, L2 + clip + Adam (+ operand store) and the next batch's rows in ONE launch (emx.h).  ``reg_ws``: 2 x
    EMX_REDUCE_WS floats (the regulariser shares by the parity of the optimizer step)."""
    _no_merged_launch(plan, "emx_mlp_wgrad_adam_local")
    if plan.aop is None:
        raise EmxError("mlp_wgrad_adam_local: the plan has no weight-gradient operand store (MlpPlan(..., train_rows=B))")
    assert sync.dtype == torch.int32 and sync.numel() >= 4 + 1024 and reg_ws.numel() >= 2 * EMX_REDUCE_WS
    nb = C.byref(next_batch) if next_batch is not None else None
    _check(load().emx_mlp_wgrad_adam_local(C.byref(plan.desc), _ptr(plan.params), plan.n_params, plan.aop.data_ptr(), plan.aop_rows, int(row_begin),
                                           int(row_end), _ptr(ws), ws.numel() if ws is not None else 0, C.byref(finish), _ptr(m), _ptr(v), _ptr(l2coef), float(lr), float(beta1), float(beta2),
                                           float(eps), float(clip if clip is not None else 0.0), _ptr(reg_ws), _ptr(plan.mirror), nb, _ptr(sync),
                                           _stream()), "emx_mlp_wgrad_adam_local")


# ---- ADC path ---------------------------------------------------------------------------------------------------
def colmean(x, out):
    B, n = x.shape
    _check(load().emx_colmean(_ptr(x), B, n, x.stride(0), _ptr(out), _stream()), "emx_colmean")


def chain_in_plane_fwd(lengths, angles, xyz):
    B, A = xyz.shape[0], xyz.shape[1]
    _check(load().emx_chain_in_plane_fwd(_ptr(lengths), _ptr(angles), angles.stride(0), B, A, _ptr(xyz), _stream()), "emx_chain_in_plane_fwd")


def chain_in_plane_bwd(lengths, angles, dxyz, dangles):
    B, A = dxyz.shape[0], dxyz.shape[1]
    _check(load().emx_chain_in_plane_bwd(_ptr(lengths), _ptr(angles), angles.stride(0), _ptr(dxyz), B, A, _ptr(dangles),
                                         dangles.stride(0), _stream()), "emx_chain_in_plane_bwd")


def dihedrals_to_cartesian_fwd(dihedrals, offset, plane, out):
    B, A = plane.shape[0], plane.shape[1]
    _check(load().emx_dihedrals_to_cartesian_fwd(_ptr(dihedrals), dihedrals.stride(0), float(offset), _ptr(plane), B, A, _ptr(out),
                                                 _stream()), "emx_dihedrals_to_cartesian_fwd")


def dihedrals_to_cartesian_bwd(dihedrals, offset, plane, dout, ddihedrals, dplane):
    B, A = plane.shape[0], plane.shape[1]
    _check(load().emx_dihedrals_to_cartesian_bwd(_ptr(dihedrals), dihedrals.stride(0), float(offset), _ptr(plane), _ptr(dout), B, A,
                                                 _ptr(ddihedrals), ddihedrals.stride(0), _ptr(dplane), _stream()),
           "emx_dihedrals_to_cartesian_bwd")


def pdist_flat_fwd(xyz, start, step, C, out):
    B, A = xyz.shape[0], xyz.shape[1]
    _check(load().emx_pdist_flat_fwd(_ptr(xyz), B, A, int(start), int(step), int(C), _ptr(out), out.stride(0), _stream()), "emx_pdist_flat_fwd")


def adc_frame_fwd(Y, periodicity, ang_cols, mean_angles, ang_pred, dih_cols, dih_pred, mean_lengths, dihedral_offset, plane, xyz, cart,
                  start, step, C, inp_pair, out_pair, split_ws=None):
    """PeriodicOutput + BackMapLayer + both PairwiseDistances blocks in one launch (emx_adc_frame_fwd).  ``ang_cols`` /
    ``dih_cols``: (sin column, cos column) in Y; ``ang_cols`` None: the angles are ``mean_angles``; ``split_ws``: also
    emx_sketchmap_split of ``inp_pair``."""
    B, A = xyz.shape[0], xyz.shape[1]
    a_sin, a_cos = ang_cols if ang_cols is not None else (-1, -1)
    if ang_cols is not None:
        assert ang_pred.is_contiguous() and ang_pred.shape == (B, A - 2)
    assert dih_pred.is_contiguous() and dih_pred.shape == (B, A - 3)
    _check(load().emx_adc_frame_fwd(_ptr(Y), Y.stride(0), B, float(periodicity), int(a_sin), int(a_cos),
                                    _ptr(mean_angles) if mean_angles is not None else None, _ptr(ang_pred) if ang_cols is not None else None,
                                    int(dih_cols[0]), int(dih_cols[1]), _ptr(dih_pred), _ptr(mean_lengths), float(dihedral_offset), A,
                                    _ptr(plane), _ptr(xyz), _ptr(cart) if cart is not None else None, int(start), int(step), int(C),
                                    _ptr(inp_pair) if inp_pair is not None else None, _ptr(out_pair), out_pair.stride(0),
                                    _ptr(split_ws) if split_ws is not None else None, _stream()),
           "emx_adc_frame_fwd")


def adc_frame_fwd_bwd(Y, periodicity, ang_cols, mean_angles, ang_pred, dih_cols, dih_pred, mean_lengths, dihedral_offset, plane, xyz, cart,
                      start, step, C, inp_pair, out_pair, split_ws, variant, coef, coef_scale, ws, ddihedrals, dplane, jobs=None, dangles=None):
    """adc_frame_fwd and adc_frame_bwd as one launch (emx_adc_frame_fwd_bwd): a workgroup back-maps its frame and runs the cartesian
    cost and the way back on what it has just written."""
    B, A = xyz.shape[0], xyz.shape[1]
    a_sin, a_cos = ang_cols if ang_cols is not None else (-1, -1)
    if ang_cols is not None:
        assert ang_pred.is_contiguous() and ang_pred.shape == (B, A - 2)
    assert dih_pred.is_contiguous() and dih_pred.shape == (B, A - 3) and out_pair.stride(0) == inp_pair.stride(0)
    arr = _cost_jobs(jobs) if jobs else None
    _check(load().emx_adc_frame_fwd_bwd(_ptr(Y), Y.stride(0), B, float(periodicity), int(a_sin), int(a_cos),
                                        _ptr(mean_angles) if mean_angles is not None else None, _ptr(ang_pred) if ang_cols is not None else None,
                                        int(dih_cols[0]), int(dih_cols[1]), _ptr(dih_pred), _ptr(mean_lengths), float(dihedral_offset), A,
                                        _ptr(plane), _ptr(xyz), _ptr(cart), int(start), int(step), int(C), _ptr(inp_pair), _ptr(out_pair),
                                        out_pair.stride(0), _ptr(split_ws) if split_ws is not None else None,
                                        VARIANTS[variant] if isinstance(variant, str) else int(variant), float(coef),
                                        _ptr(coef_scale) if coef_scale is not None else None, _ptr(ws), _ptr(ddihedrals), ddihedrals.stride(0),
                                        _ptr(dplane), _ptr(dangles) if ang_cols is not None else None,
                                        dangles.stride(0) if (ang_cols is not None and dangles is not None) else 0,
                                        _void_p(arr) if arr is not None else None, len(jobs) if jobs else 0, _stream()),
           "emx_adc_frame_fwd_bwd")


def adc_frame_bwd(out_pair, inp_pair, variant, coef, coef_scale, ws, xyz, dihedrals, offset, plane, start, step, C, ddihedrals, dplane,
                  jobs=None, mean_lengths=None, angles=None, dangles=None):
    """Cartesian cost + backward of the pair distances + backward of dihedrals_to_cartesian in one launch (emx_adc_frame_bwd);
    ``angles`` (+ ``mean_lengths``, ``dangles``): chain_in_plane's backward pass behind it; ``jobs``: the job dicts of
    ``periodic_cost_batch`` behind that, all in the same launch."""
    B, A = xyz.shape[0], xyz.shape[1]
    assert out_pair.stride(0) == inp_pair.stride(0) and dihedrals.is_contiguous() and (angles is None or angles.is_contiguous())
    arr = _cost_jobs(jobs) if jobs else None
    _check(load().emx_adc_frame_bwd(_ptr(out_pair), _ptr(inp_pair), out_pair.stride(0), VARIANTS[variant] if isinstance(variant, str) else int(variant),
                                    float(coef), _ptr(coef_scale) if coef_scale is not None else None, _ptr(ws), _ptr(xyz), _ptr(dihedrals),
                                    float(offset), _ptr(plane), B, A, int(start), int(step), int(C), _ptr(ddihedrals), ddihedrals.stride(0),
                                    _ptr(dplane), _ptr(mean_lengths) if angles is not None else None, _ptr(angles) if angles is not None else None,
                                    _ptr(dangles) if angles is not None else None, dangles.stride(0) if angles is not None else 0,
                                    _void_p(arr) if arr is not None else None, len(jobs) if jobs else 0, _stream()),
           "emx_adc_frame_bwd")


def pdist_flat_bwd(xyz, dout, start, step, C, dxyz, accumulate=False):
    B, A = xyz.shape[0], xyz.shape[1]
    _check(load().emx_pdist_flat_bwd(_ptr(xyz), _ptr(dout), B, A, int(start), int(step), int(C), dout.stride(0), _ptr(dxyz),
                                     int(bool(accumulate)), _stream()), "emx_pdist_flat_bwd")


def soft_start_scale(step, a, b, c, out):
    _check(load().emx_soft_start_scale(_ptr(step), float(a), float(b), float(c), _ptr(out), _stream()), "emx_soft_start_scale")


def adc_batch_head(feat_data, cart_data, idx, cursor, B, feat, cart, parts, periodicity, xin, step, soft_a, soft_b, scale_c, cart_scale,
                   feat_mean=None):
    """Gather of the feature and Cartesian rows, sin/cos of the angular blocks ``parts`` = [(off, n, col), ...], the
    soft-start scale and (``feat_mean``) the column means of the gathered feature rows in one launch (emx_adc_batch_head)."""
    N, F = feat_data.shape
    C3 = cart_data.shape[1]
    n = len(parts)
    arr = lambda k: (C.c_int * n)(*[int(p[k]) for p in parts])
    _check(load().emx_adc_batch_head(_ptr(feat_data), _ptr(cart_data), N, F, C3, _ptr(idx), _ptr(cursor), int(B), _ptr(feat), _ptr(cart),
                                     arr(0), arr(1), arr(2), n, float(periodicity), _ptr(xin), xin.stride(0), _ptr(step), float(soft_a),
                                     float(soft_b), float(scale_c), _ptr(cart_scale), _ptr(feat_mean) if feat_mean is not None else None,
                                     _stream()), "emx_adc_batch_head")


def adc_head_describe(feat_data, cart_data, B, feat, cart, parts, periodicity, xin, step, soft_a, soft_b, scale_c, cart_scale, feat_mean,
                      n_batches) -> torch.Tensor:
    """The device record of ``adc_batch_head``'s description (emx_adc_head_describe) for ``make_next_adc_head``."""
    N, F = feat_data.shape
    C3 = cart_data.shape[1]
    n = len(parts)
    arr = lambda k: (C.c_int * n)(*[int(p[k]) for p in parts])
    # (no fill kernel: it would run on torch's current stream, possibly BEHIND the blocking copy of emx_adc_head_describe)
    rec = torch.empty(int(load().emx_adc_head_bytes()), dtype=torch.uint8, device=feat_data.device)
    torch.cuda.current_stream(feat_data.device).synchronize()
    _check(load().emx_adc_head_describe(_ptr(feat_data), _ptr(cart_data), N, F, C3, int(B), _ptr(feat), _ptr(cart), arr(0), arr(1), arr(2), n,
                                        float(periodicity), _ptr(xin), xin.stride(0), _ptr(step), float(soft_a), float(soft_b), float(scale_c),
                                        _ptr(cart_scale), _ptr(feat_mean) if feat_mean is not None else None, int(n_batches), _ptr(rec)),
           "emx_adc_head_describe")
    rec.keep = (feat_data, cart_data, feat, cart, xin, step, cart_scale, feat_mean)
    return rec


def allatom_mask_words(A):
    return int(load().emx_allatom_mask_words(int(A)))


def allatom_backmap(xyz0, quads, far_mask, targets, out):
    """xyz0 [A, 3] (one template for all frames) or [n, A, 3]; quads int32 [J, 4]; far_mask int32 [J, W, 64]; targets [n, J]."""
    n, A = out.shape[0], out.shape[1]
    J = quads.shape[0]
    stride = 0 if xyz0.dim() == 2 else xyz0.stride(0)
    _check(load().emx_allatom_backmap(_ptr(xyz0), stride, n, A, _ptr(quads), _ptr(far_mask), J, _ptr(targets), targets.stride(0) if J else 0,
                                      _ptr(out), _stream()), "emx_allatom_backmap")


def backbone_features(xyz, distances, angles, dihedrals):
    N, A = xyz.shape[0], xyz.shape[1]
    _check(load().emx_backbone_features(_ptr(xyz), N, A, _ptr(distances), _ptr(angles), _ptr(dihedrals), _stream()), "emx_backbone_features")
