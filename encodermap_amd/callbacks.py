"""Training callbacks with the cadence of the reference (encodermap/callbacks/callbacks.py:112-211,272-330,
519-529): a private step counter that starts at 0, ``on_checkpoint_step`` when ``counter % checkpoint_step
== 0`` and ``on_summary_step`` when ``counter % summary_step == 0`` (checkpoint first).

The engine enqueues steps asynchronously and only synchronises at the steps where some callback is due, so
``logs["loss"]`` is available exactly there.  A callback that is not derived from
``EncoderMapBaseCallback`` but has ``on_train_batch_end`` is called after every step (slow path: one
synchronisation per step, like Keras).
"""
from __future__ import annotations

import sys
from typing import Optional


class EncoderMapBaseCallback:
    def __init__(self, parameters=None) -> None:
        if parameters is None:
            from .parameters import Parameters

            parameters = Parameters()
        self.p = parameters
        self.steps_counter = 0
        self.model = None  # set by the autoencoder class (the object that owns save())

    def set_model(self, model) -> None:
        self.model = model

    # steps (1-based counter values) in (lo, hi] at which this callback wants control
    def due_steps(self, lo: int, hi: int) -> list:
        out = set()
        for period in (self.p.checkpoint_step, self.p.summary_step):
            if period and period > 0:
                first = (lo // period + 1) * period
                out.update(range(first, hi + 1, period))
        return sorted(out)

    def on_train_batch_end(self, batch: int, logs: Optional[dict] = None) -> None:
        self.steps_counter += 1
        if self.steps_counter % self.p.checkpoint_step == 0:
            self.on_checkpoint_step(self.steps_counter, logs=logs)
        if self.steps_counter % self.p.summary_step == 0:
            self.on_summary_step(self.steps_counter, logs=logs)

    def on_summary_step(self, step: int, logs: Optional[dict] = None) -> None:
        pass

    def on_checkpoint_step(self, step: int, logs: Optional[dict] = None) -> None:
        pass

    def on_train_begin(self, logs: Optional[dict] = None) -> None:
        pass

    def on_train_end(self, logs: Optional[dict] = None) -> None:
        pass


class ProgressBar(EncoderMapBaseCallback):
    """Loss read-out every ``summary_step`` steps (reference: tqdm bar, callbacks.py:272-330).  Silent unless
    ``enabled`` -- a read-out costs one device synchronisation."""

    def __init__(self, parameters=None, enabled: bool = False) -> None:
        super().__init__(parameters)
        self.enabled = enabled

    def due_steps(self, lo: int, hi: int) -> list:
        return super().due_steps(lo, hi) if self.enabled else []

    def on_summary_step(self, step: int, logs: Optional[dict] = None) -> None:
        if self.enabled:
            loss = (logs or {}).get("loss", float("nan"))
            total = self.p.n_steps - self.p.current_training_step
            sys.stderr.write(f"\r{step}/{total} steps, loss {loss:.6g}")
            sys.stderr.flush()

    def on_train_end(self, logs: Optional[dict] = None) -> None:
        if self.enabled:
            sys.stderr.write("\n")


class CheckpointSaver(EncoderMapBaseCallback):
    """Writes ``saved_model_{step}`` every ``checkpoint_step`` steps (callbacks.py:519-529)."""

    def due_steps(self, lo: int, hi: int) -> list:
        period = self.p.checkpoint_step
        if not period or period <= 0:
            return []
        return list(range((lo // period + 1) * period, hi + 1, period))

    def on_checkpoint_step(self, step: int, logs: Optional[dict] = None) -> None:
        if self.model is not None:
            self.model.save(step=step, print_message=False)


class EarlyStop(EncoderMapBaseCallback):
    """Stops when the loss has not changed by more than ``min_delta`` for ``patience`` summary steps
    (callbacks.py:219-269, simplified to the loss criterion)."""

    def __init__(self, parameters=None, patience: int = 0, min_delta: float = 1e-5) -> None:
        super().__init__(parameters)
        self.patience, self.min_delta = patience, min_delta
        self.wait, self.last = 0, None
        self.stop_training = False

    def on_summary_step(self, step: int, logs: Optional[dict] = None) -> None:
        loss = (logs or {}).get("loss")
        if loss is None:
            return
        if self.last is not None and abs(self.last - loss) < self.min_delta:
            self.wait += 1
            if self.wait >= self.patience:
                self.stop_training = True
        else:
            self.wait = 0
        self.last = loss


class History:
    """Stand-in for ``tf.keras.callbacks.History``: ``.history['loss']`` holds one float per step."""

    def __init__(self, losses, terms=None, params=None, metrics=None) -> None:
        self.history = {"loss": [float(v) for v in losses]}
        if terms:
            for k, v in terms.items():
                self.history[f"{k}_cost"] = [float(x) for x in v]
        if metrics:
            for k, v in metrics.items():   # Keras logs every metric's result() under the metric's name
                self.history[k] = [float(x) for x in v]
        self.epoch = list(range(len(self.history["loss"])))
        self.params = params or {}


# ---------------------------------------------------------------------------------------------------------------------
# Training metrics of the ADC model (encodermap/callbacks/metrics.py:470-581 of the reference).  During training the
# engine evaluates them inside the captured step (ADCParameters.track_clashes / track_RMSD); the classes below keep the
# reference's stand-alone interface -- update_state(y_true, y_pred) / result() -- on the same HIP kernels.
# ---------------------------------------------------------------------------------------------------------------------
def _frames(y):
    """[B, A, 3] float32 CUDA tensor from an array, a tensor, or the model's output tuple (frames are entry 2)."""
    import numpy as np
    import torch

    if isinstance(y, (list, tuple)):
        y = y[2]
    t = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(y, dtype=np.float32)))
    return t.to("cuda", torch.float32).contiguous()


class ADCClashMetric:
    """Number of atom pairs closer than 0.1 nm / 1 Angstrom in the reconstructed backbone, averaged over the batch
    (metrics.py:470-525; the mean of the int64 counts is truncated like ``tf.reduce_mean``)."""

    def __init__(self, distance_unit: str, name: str = "ADCClashMetric", parameters=None, **kwargs) -> None:
        if distance_unit == "nm":
            self.clash_distance = 0.1
        elif distance_unit == "ang":
            self.clash_distance = 1
        else:
            raise Exception(f"Argument `distance_unit` must be either 'nm' or 'ang'. You provided {distance_unit=}.")
        self.name = name
        self.p = parameters
        self.clashes = 0

    def update_state(self, y_true, y_pred, sample_weight=None) -> None:
        import torch

        from . import _lib

        _lib.require_gpu()
        xyz = _frames(y_pred)
        counts = torch.zeros(xyz.shape[0], dtype=torch.int64, device=xyz.device)
        _lib.clash_count(xyz, self.clash_distance, counts)
        self.counts = counts.cpu().numpy()
        self.clashes = int(self.counts.sum() // len(self.counts))

    def result(self):
        return self.clashes


class ADCRMSDMetric:
    """Weighted Kabsch RMSD of the output frames against the input frames, one value per batch row (metrics.py:528-581)."""

    WEIGHTS = (14.0067, 24.305, 24.305)   # metrics.py:63

    def __init__(self, name: str = "ADCRMSDMetric", parameters=None, **kwargs) -> None:
        self.name = name
        self.p = parameters
        self.rmsd = None

    def update_state(self, y_true, y_pred, sample_weight=None) -> None:
        import torch

        from . import _lib

        _lib.require_gpu()
        a, b = _frames(y_true), _frames(y_pred)
        out = torch.zeros(a.shape[0], dtype=torch.float32, device=a.device)
        _lib.kabsch_rmsd(a, b, self.WEIGHTS, True, out)
        self.rmsd = out.cpu().numpy()

    def result(self):
        return self.rmsd
