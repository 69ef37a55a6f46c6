"""User-facing classes ``Autoencoder``, ``EncoderMap``, ``DihedralEncoderMap`` -- same constructor arguments,
methods, side effects and error behaviour as the reference's ``encodermap/autoencoder/autoencoder.py``
(``Autoencoder`` :573-1229, ``EncoderMap`` :1232-1307, ``DihedralEncoderMap`` :1310-1400), backed by
``engine.SequentialEngine`` (hand-written HIP kernels on an MI355X, include/emx.h) instead of Keras.

Not carried over (out of the hot-path scope): sparse/NaN input, ``tf.data.Dataset`` input,
TensorBoard, user-supplied Keras models and loss closures -- each raises ``NotImplementedError`` with a message
instead of silently doing something else.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Sequence, Union

import numpy as np

from . import checkpoint as ckpt
from .callbacks import CheckpointSaver, EncoderMapBaseCallback, History, ProgressBar
from .misc import create_n_cube
from .parameters import ADCParameters, Parameters

INF = float("inf")


class _SubModel:
    """What ``model.encoder_model`` / ``model.decoder_model`` expose in the reference and user code touches."""

    def __init__(self, engine, layers, input_width, fn):
        self._engine, self.layers_spec, self._fn = engine, layers, fn
        self.input_shape = (None, input_width)
        self.output_shape = (None, layers[-1].n_out)

    def __call__(self, x, training: bool = False):
        return self._fn(np.asarray(x))

    @property
    def layers(self):
        return list(self.layers_spec)

    def get_weights(self):
        w = self._engine.get_weights()
        first = self._engine.layout.layers.index(self.layers_spec[0])
        return w[2 * first: 2 * (first + len(self.layers_spec))]


class ModelView:
    """Stand-in for the Keras ``SequentialModel`` object (models/models.py:3099-3401): ``encoder(x)`` applies the
    sin/cos transform for periodic data, ``encoder_model(x)`` expects the already transformed input,
    ``decoder(z)`` returns angles (atan2 of the halves) for periodic data."""

    def __init__(self, engine):
        self.engine = engine
        lay = engine.layout
        self.encoder_model = _SubModel(engine, lay.encoder, lay.net_dim, lambda x: engine.encode(x, already_transformed=True))
        self.decoder_model = _SubModel(engine, lay.decoder, lay.latent_dim, self._decoder_raw)

    def _decoder_raw(self, z):
        # decoder_model in the reference returns the raw last Dense layer (before atan2)
        return self.engine.decode_raw(np.asarray(z))

    def encoder(self, x, training: bool = False):
        return self.engine.encode(np.asarray(x))

    def decoder(self, z, training: bool = False):
        return self.engine.decode(np.asarray(z))

    def __call__(self, x, training: bool = False):
        return self.decoder(self.encoder(x))

    @property
    def layers(self):
        return list(self.engine.layout.layers)

    def get_weights(self):
        return self.engine.get_weights()

    def set_weights(self, arrays):
        self.engine.set_weights(arrays)


class Autoencoder:
    """Main autoencoder class: auto cost + regularisation cost + center cost (autoencoder.py:573-1229).

    Args (as in the reference): ``parameters`` (``Parameters`` or None for defaults), ``train_data`` (None for the
    4-d hypercube test data, or ``np.ndarray [N, D]``), ``model`` (None, or a list ``[W0, b0, W1, b1, ...]`` of
    kernels/biases in Keras ``get_weights`` order to start from), ``read_only`` (do not write to disk), ``sparse``.
    Engine-only keyword arguments: ``device``, ``dp`` (``engine.DataParallel``), ``use_graph``, ``progress``.
    """

    _with_distance = False
    _loss_names = ("auto_loss", "regularization_loss", "center_loss")

    def __init__(self, parameters=None, train_data: Optional[np.ndarray] = None, model=None, read_only: bool = False,
                 sparse: bool = False, *, device: Optional[str] = None, dp=None, use_graph: bool = True,
                 progress: bool = False) -> None:
        self.p = Parameters() if parameters is None else parameters
        self.read_only = read_only
        if not self.read_only:
            self.p.write_summary = True
            self.p.save()
            self._print_save_message()
        if self.p.tensorboard:
            raise NotImplementedError("TensorBoard output is not part of the HIP engine (SURVEY.md section 2: out of scope).")
        if sparse:
            raise NotImplementedError("sparse input is not part of the HIP engine (SURVEY.md section 2: out of scope).")
        self.sparse = False
        self._engine_kw = dict(device=device or "cuda:0", dp=dp, use_graph=use_graph)
        self._progress = progress
        self.set_train_data(train_data)
        self._weights = None
        if model is not None:
            if isinstance(model, (list, tuple)):
                self._weights = [np.asarray(a, np.float32) for a in model]
            else:
                raise NotImplementedError(
                    "`model` must be None or a list of kernel/bias arrays; custom Keras models cannot run on the HIP engine.")
        self.model = self._get_model()
        self._setup_callbacks()
        self.loss = self._get_loss()
        self.optimizer = {"name": "Adam", "learning_rate": self.p.learning_rate, "clipvalue": 1.0,
                          "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7}

    # -- construction helpers ---------------------------------------------------------------------
    def _print_save_message(self) -> None:
        print(f"Output files are saved to {self.p.main_path} as defined in 'main_path' in the parameters.")

    def _print_missing_data(self) -> None:
        print(
            f"This model was reloaded from disk, but not yet provided with train data. Use the `set_train_data()` method "
            f"to provide the train data and call `train()` again to train the model. Alternatively, you could directly "
            f"provide the train data, when reloading by calling the `{self.__class__.__name__}.from_checkpoint()` "
            f"constructor with the `train_data` argument. Expected shape = {self.model.encoder_model.input_shape[1]}, "
            f"received shape = {self._data_shape} {self._using_hypercube=} {self.p.using_hypercube=}")

    def set_train_data(self, data: Optional[np.ndarray]) -> None:
        """(Re)sets the train data (autoencoder.py:788-866)."""
        self._using_hypercube = False
        if data is None:
            self._using_hypercube = True
            self.p.using_hypercube = True
            self.train_data = create_n_cube(4, seed=self.p.seed)[0].astype("float32")
            self.p.periodicity = float("inf")
        elif isinstance(data, np.ndarray):
            if np.any(np.isnan(data)):
                raise NotImplementedError("Input contains nans; the sparse network is not part of the HIP engine.")
            self.train_data = data.astype("float32")
            if np.any(data > self.p.periodicity):
                raise Exception(
                    "There seems to be an error regarding the periodicity of your data. The chosen periodicity is "
                    f"{self.p.periodicity}, but there are datapoints outwards of this range: {data.max()}")
        else:
            raise TypeError(f"train_data must be `None` or `np.ndarray`. You supplied {type(data)}.")
        if self.train_data.ndim != 2:
            raise Exception(f"train_data must be 2-dimensional [n_points, n_features], got shape {self.train_data.shape}")
        self._data_shape = self.train_data.shape
        self.dataset = self.train_data  # the reference exposes a tf.data.Dataset here
        eng = getattr(self, "_engine", None)
        if eng is not None and self.train_data.shape[1] == eng.D:
            eng.set_data(self.train_data)
            self._data_missing = False   # a model reloaded without data (from_checkpoint) can train from here on

    def _get_model(self) -> ModelView:
        from .engine import SequentialEngine

        cap = max(int(self.p.n_steps) - int(self.p.current_training_step), 1)
        self._engine = SequentialEngine(self.train_data.shape[1], self.p, with_distance=self._with_distance,
                                        weights=self._weights, history_capacity=cap, **self._engine_kw)
        self._engine.set_data(self.train_data)
        return ModelView(self._engine)

    def _setup_callbacks(self) -> None:
        self.metrics = []
        self.callbacks = [ProgressBar(parameters=self.p, enabled=self._progress)]
        if not self.read_only:
            cs = CheckpointSaver(self.p)
            cs.set_model(self)
            self.callbacks.append(cs)
        self.tensorboard_write_bool = None

    def _get_loss(self):
        loss = self.p.loss
        if loss == "emap_cost":
            return list(self._loss_names)
        if loss in ("reconstruction_loss", "mse"):
            # (autoencoder.py:948-958: the plain mean-square reconstruction error, nothing else -- engine._PlainReconstruction)
            return ["reconstruction_loss"]
        raise ValueError(f"loss argument needs to be `reconstruction_loss`, `mse` or `emap_cost`. You provided '{loss}'.")

    @classmethod
    def from_checkpoint(cls, checkpoint_path: Union[str, Path], train_data: Optional[np.ndarray] = None, sparse: bool = False,
                        use_previous_model: bool = False, compat: bool = False, **engine_kw):
        """Rebuilds the class from ``saved_model_*.emx.npz`` + ``parameters.json`` (autoencoder.py:1236-1277)."""
        file = ckpt.resolve(checkpoint_path)
        parameter_file = file.parent / "parameters.json"
        if not parameter_file.is_file():
            raise Exception(f"There was no parameters.json file in the directory. {parameter_file.parent}")
        pcls = ADCParameters if "cartesian" in parameter_file.read_text() else Parameters
        p = pcls.from_file(parameter_file)
        p = ckpt.reconcile_step(file, p, use_previous_model)
        payload = ckpt.load_checkpoint(file)
        n_layers = sum(1 for k in payload if k.endswith("/kernel"))
        arrays = []
        for i in range(n_layers):
            stem = "dense" if i == 0 else f"dense_{i}"
            arrays += [payload[f"{stem}/kernel"], payload[f"{stem}/bias"]]
        was_cube = bool(p.using_hypercube)
        if train_data is None and not was_cube:
            # build around placeholder data of the right width; train() will ask for real data
            width = arrays[0].shape[0] // (2 if p.periodicity < INF else 1)
            new = cls(p, np.zeros((max(p.batch_size, 1), width), np.float32), model=arrays, read_only=False, **engine_kw)
            new._data_missing = True
        else:
            new = cls(p, train_data, model=arrays, read_only=False, **engine_kw)
        new.p.using_hypercube = was_cube
        if "adam/m" in payload and payload["adam/m"].shape == (new._engine.layout.n_params,):
            new._engine.load_state_dict({"m": payload["adam/m"], "v": payload["adam/v"], "step": int(payload["adam/iterations"])})
        return new

    # -- properties -----------------------------------------------------------------------------------
    @property
    def encoder(self):
        return self.model.encoder

    @property
    def decoder(self):
        return self.model.decoder

    # -- training -------------------------------------------------------------------------------------
    def train(self) -> Optional[History]:
        """Trains for ``n_steps - current_training_step`` steps (autoencoder.py:967-1029).  Returns a History-like
        object (``.history['loss']``), or None if there is nothing to do."""
        name = self.__class__.__name__
        if self.p.current_training_step >= self.p.n_steps:
            print(f"This {name} instance has already been trained for {self.p.current_training_step} steps. Increase the "
                  f"training steps by calling `{name}.p.n_steps += new_steps` and then call `{name}.train()` again.")
            return None
        if getattr(self, "_data_missing", False) or self._using_hypercube != self.p.using_hypercube:
            return self._print_missing_data()
        if self._data_shape[1] != self._engine.D:
            return self._print_missing_data()
        if self.p.training == "custom":
            raise NotImplementedError()
        if self.p.training != "auto":
            raise ValueError(f"training argument needs to be `auto` or `custom`. You provided '{self.p.training}'.")
        eng = self._engine
        total = int(self.p.n_steps) - int(self.p.current_training_step)
        start_hist = eng.hstep_host()
        eng.reserve_history(total)
        for cb in self.callbacks:
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin()
        generic = [cb for cb in self.callbacks if not isinstance(cb, EncoderMapBaseCallback) and hasattr(cb, "on_train_batch_end")]
        based = [cb for cb in self.callbacks if isinstance(cb, EncoderMapBaseCallback)]
        done = 0
        stop = False
        while done < total and not stop:
            # next step (relative, 1-based) at which some callback is due
            nxt = total
            if generic:
                nxt = done + 1
            for cb in based:
                due = cb.due_steps(cb.steps_counter, cb.steps_counter + (total - done))
                if due:
                    nxt = min(nxt, done + (due[0] - cb.steps_counter))
            k = nxt - done
            eng.run_steps(k)
            done = nxt
            need_logs = bool(generic) or any(cb.due_steps(cb.steps_counter, cb.steps_counter + k) for cb in based)
            logs = None
            if need_logs:
                loss = eng.last_loss()
                self._assert_finite(loss)
                logs = {"loss": loss}
            for cb in based:
                cb.steps_counter += k
                if cb.p.checkpoint_step and cb.steps_counter % cb.p.checkpoint_step == 0 and cb.due_steps(cb.steps_counter - 1, cb.steps_counter):
                    cb.on_checkpoint_step(cb.steps_counter, logs=logs)
                if cb.p.summary_step and cb.steps_counter % cb.p.summary_step == 0 and cb.due_steps(cb.steps_counter - 1, cb.steps_counter):
                    cb.on_summary_step(cb.steps_counter, logs=logs)
                stop = stop or bool(getattr(cb, "stop_training", False))
            for cb in generic:
                cb.on_train_batch_end(done - 1, logs)
                stop = stop or bool(getattr(cb, "stop_training", False))
        losses = eng.losses(start_hist)
        terms = eng.loss_terms(start_hist)
        for v in losses:
            self._assert_finite(float(v))
        for cb in self.callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end()
        self.p.current_training_step += done
        if not self.read_only:
            self.p.save()
        self.save()
        metrics = {}
        if hasattr(eng, "metric_history"):
            metrics = {k: v[start_hist:] for k, v in eng.metric_history().items()}
        return History(losses, terms, params={"epochs": total, "steps": 1}, metrics=metrics)

    @staticmethod
    def _assert_finite(loss: float) -> None:
        if not np.isfinite(loss):
            raise Exception("Cost became infinite or NaN.")   # tf.debugging.assert_all_finite in every loss closure

    def add_loss(self, loss) -> None:
        raise NotImplementedError("custom loss closures operate on Keras models; the HIP engine runs the built-in EncoderMap costs only.")

    def add_callback(self, callback) -> None:
        """Adds a callback class (instantiated with the parameters) or instance (autoencoder.py:1038-1043)."""
        try:
            cb = callback(self.p)
        except TypeError:
            cb = callback
        if hasattr(cb, "set_model"):
            cb.set_model(self)
        self.callbacks.append(cb)

    def add_metric(self, metric) -> None:
        self.metrics.append(metric(self.p))

    # -- inference ------------------------------------------------------------------------------------
    def encode(self, data: Optional[np.ndarray] = None) -> np.ndarray:
        """Projects ``data`` (default: the train data) to the latent space (autoencoder.py:1110-1152)."""
        if data is None:
            eng = getattr(self, "_engine", None)
            if eng is not None and eng.data is not None and tuple(eng.data.shape) == tuple(np.shape(self.train_data)):
                return eng.encode_resident()      # the training set already lives in HBM: only the latent rows come back
            data = self.train_data
        if isinstance(data, (list, tuple)):
            raise Exception("a sequence of arrays is only valid for AngleDihedralCartesianEncoderMap")
        data = np.asarray(data)
        width = self.model.encoder_model.input_shape[1]
        if data.shape[1] * 2 == width:
            return self.model.encoder(data)
        if data.shape[1] == width:
            return self.model.encoder_model(data)
        msg = f"The shape of the provided data {data.shape=} does not match the expected shape {self.model.encoder_model.input_shape=}."
        if self.p.periodicity < INF:
            msg += f" Not even considering the periodicity of {self.p.periodicity}."
        raise Exception(msg)

    def generate(self, data: np.ndarray) -> np.ndarray:
        """Decoder output for latent points (autoencoder.py:1154-1171; the reference returns a tf.Tensor)."""
        return self.model.decoder(data)

    def decode(self, data: np.ndarray) -> np.ndarray:
        return self.model.decoder(data)

    # -- persistence ----------------------------------------------------------------------------------
    def save(self, step: Optional[int] = None, print_message: bool = True) -> Optional[Path]:
        """Writes ``saved_model_{step|time}.emx.npz`` into ``p.main_path`` (autoencoder.py:1197-1224)."""
        name = self.__class__.__name__
        if self.read_only:
            print(f"This {name} is set to read_only. Set `{name}.read_only=False` to save the current state of the model.")
            return None
        fname = ckpt.checkpoint_name(self.p.main_path, step)
        if print_message:
            print(f"Saving the model to {fname}. Use `em.{name}.from_checkpoint('{self.p.main_path}')` to load the most "
                  f"recent model, or `em.{name}.from_checkpoint('{fname}')` to load the model with specific weights..")
        ckpt.save_checkpoint(fname, self._engine.get_weights(), self._engine.state_dict(), name)
        return fname.parent

    def close(self) -> None:
        """Frees the device buffers."""
        self._engine = None
        self.model = None
        import torch

        if torch.cuda.is_available():
            torch.cuda.empty_cache()


class EncoderMap(Autoencoder):
    """Autoencoder + sketch-map distance cost (autoencoder.py:1232-1307)."""

    _with_distance = True
    _loss_names = ("auto_loss", "regularization_loss", "center_loss", "distance_loss")


class DihedralEncoderMap(EncoderMap):
    """EncoderMap on backbone dihedrals (autoencoder.py:1310-1400).  ``generate(z, top)`` rebuilds molecular
    conformations through MDAnalysis in the reference -- outside the hot path; ``decode`` gives the dihedrals."""

    def generate(self, data: np.ndarray, top=None):
        if top is not None:
            raise NotImplementedError("back-mapping dihedrals onto a topology needs MDAnalysis (out of scope); use decode().")
        return self.model.decoder(data)
