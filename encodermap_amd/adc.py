"""``AngleDihedralCartesianEncoderMap`` -- the ADC autoencoder of the reference
(encodermap/autoencoder/autoencoder.py:1403-2576) on the HIP engine (``adc_engine.ADCEngine``).

The reference consumes ``TrajEnsemble`` objects (MD file I/O + featurisation, out of scope); what the training path
actually reads from them is the dict ``trajs.CVs`` with the float32 arrays ``central_angles [N, A-2]``,
``central_dihedrals [N, A-3]``, ``central_cartesians [N, A, 3]``, ``central_distances [N, A-1]`` and
``side_dihedrals [N, S]`` (autoencoder.py:1613-1621).  This class accepts exactly that: an object with a ``CVs``
attribute, the dict itself, or the 4/5-tuple in the reference's dataset order (angles, dihedrals, cartesians,
distances[, side_dihedrals]).
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional, Sequence, Union

import numpy as np

from . import checkpoint as ckpt
from .autoencoder import Autoencoder, ModelView
from .parameters import ADCParameters

KEYS = ("central_angles", "central_dihedrals", "central_cartesians", "central_distances", "side_dihedrals")


def _as_cv_dict(trajs) -> dict:
    if trajs is None:
        raise Exception("AngleDihedralCartesianEncoderMap needs input data: pass `trajs` (object with .CVs / dict) or `dataset` (tuple)")
    if hasattr(trajs, "CVs"):
        trajs = trajs.CVs
    if isinstance(trajs, dict):
        missing = [k for k in KEYS[:4] if k not in trajs]
        if missing:
            raise Exception(f"Could not load CVs. Should be {set(KEYS)}, but currently only {set(trajs.keys())} are loaded.")
        return {k: np.asarray(getattr(trajs[k], "values", trajs[k])).squeeze() if k != "central_cartesians" else np.asarray(
            getattr(trajs[k], "values", trajs[k])) for k in KEYS if k in trajs}
    if isinstance(trajs, (tuple, list)) and len(trajs) in (4, 5):
        return {k: np.asarray(v) for k, v in zip(KEYS, trajs)}
    raise TypeError(f"cannot interpret input of type {type(trajs)}: expected an object with `.CVs`, a dict or a 4/5-tuple of arrays")


class _ADCModelView(ModelView):
    def __init__(self, engine):
        self.engine = engine

        class _Sub:
            pass

        self.encoder_model = _Sub()
        self.encoder_model.input_shape = [(None, n) for _, n, _ in engine.parts]
        self.decoder_model = _Sub()
        self.decoder_model.input_shape = (None, engine.L)

    def encoder(self, x, training: bool = False):
        return self.engine.encode(list(x) if isinstance(x, (list, tuple)) else [x])

    def decoder(self, z, training: bool = False):
        out = self.engine.decode(np.asarray(z))
        return out if len(out) > 1 else out[0]


class AngleDihedralCartesianEncoderMap(Autoencoder):
    """Autoencoder on backbone angles/dihedrals with a cartesian (pair-distance) cost through the dihedral->Cartesian
    back-mapping.  Arguments as in the reference (autoencoder.py:1460-1470); ``ensemble``, ``use_dataset_when_possible``
    and ``deterministic`` are accepted for signature compatibility."""

    _loss_names = ("dihedral_loss", "angle_loss", "cartesian_loss", "distance_loss", "cartesian_distance_loss",
                   "center_loss", "regularization_loss", "side_dihedral_loss")

    def __init__(self, trajs=None, parameters: Optional[ADCParameters] = None, model=None, read_only: bool = False, dataset=None,
                 ensemble: bool = False, use_dataset_when_possible: bool = True, deterministic: bool = False, *,
                 device: Optional[str] = None, dp=None, use_graph: bool = True, progress: bool = False) -> None:
        self.p = ADCParameters() if parameters is None else parameters
        if not isinstance(self.p, ADCParameters):
            raise TypeError("`parameters` must be an instance of ADCParameters")
        self.read_only = read_only
        if not self.read_only:
            self.p.write_summary = True
            self.p.save()
            self._print_save_message()
        if self.p.tensorboard:
            raise NotImplementedError("TensorBoard output is not part of the HIP engine (SURVEY.md section 2: out of scope).")
        self.trajs = trajs
        cv = _as_cv_dict(dataset if dataset is not None else trajs)
        if self.p.use_sidechains and "side_dihedrals" not in cv:
            raise Exception("use_sidechains=True needs `side_dihedrals`")
        self.inp_CV_data = {k: np.asarray(v, np.float32) for k, v in cv.items()}
        cart = self.inp_CV_data["central_cartesians"]
        N = self.inp_CV_data["central_dihedrals"].shape[0]
        if cart.ndim == 2:
            cart = cart.reshape(N, -1, 3)
            self.inp_CV_data["central_cartesians"] = cart
        self._n_atoms = cart.shape[1]
        self._n_side = self.inp_CV_data["side_dihedrals"].shape[1] if "side_dihedrals" in self.inp_CV_data else 0
        self.sparse = False
        self._engine_kw = dict(device=device or "cuda:0", dp=dp, use_graph=use_graph)
        self._progress = progress
        self._weights = None
        if model is not None:
            if isinstance(model, (list, tuple)):
                self._weights = [np.asarray(a, np.float32) for a in model]
            else:
                raise NotImplementedError("`model` must be None or a list of kernel/bias arrays")
        self.model = self._get_model()
        self.train_data = self._encoder_arrays(self.inp_CV_data)
        self.dataset = tuple(self.inp_CV_data[k] for k in KEYS if k in self.inp_CV_data)
        self._using_hypercube = False
        self._data_shape = (N, self._engine.D)
        self._setup_callbacks()
        self.loss = list(self._loss_names)
        self.optimizer = {"name": "Adam", "learning_rate": self.p.learning_rate, "clipvalue": 1.0,
                          "beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7}

    def _encoder_arrays(self, cv: dict) -> list:
        out = []
        if self.p.use_backbone_angles:
            out.append(cv["central_angles"])
        out.append(cv["central_dihedrals"])
        if self.p.use_sidechains:
            out.append(cv["side_dihedrals"])
        return out

    def _get_model(self):
        from .adc_engine import ADCEngine

        cap = max(int(self.p.n_steps) - int(self.p.current_training_step), 1)
        self._engine = ADCEngine(self._n_atoms, self._n_side, self.p, weights=self._weights, history_capacity=cap, **self._engine_kw)
        d = self.inp_CV_data
        self._engine.set_data(d["central_angles"], d["central_dihedrals"], d["central_cartesians"], d["central_distances"],
                              d.get("side_dihedrals"))
        return _ADCModelView(self._engine)

    def set_train_data(self, data) -> None:
        cv = _as_cv_dict(data)
        self.inp_CV_data = {k: np.asarray(v, np.float32) for k, v in cv.items()}
        d = self.inp_CV_data
        self._engine.set_data(d["central_angles"], d["central_dihedrals"], d["central_cartesians"], d["central_distances"],
                              d.get("side_dihedrals"))
        self._data_missing = False

    # -- training -----------------------------------------------------------------------------------------
    def train(self):
        """train_for_references first while all cost references are still 1 (autoencoder.py:1776-1779), then the loop."""
        if all(r == 1 for r in (self.p.cartesian_cost_reference, self.p.angle_cost_reference, self.p.dihedral_cost_reference)):
            self.train_for_references()
        return super().train()

    def train_for_references(self, subsample: int = 100, maxiter: int = 500) -> dict:
        """Costs of predicting the data-set means, stored as ``*_cost_reference`` (autoencoder.py:1816-1937)."""
        d = self.inp_CV_data
        N = d["central_dihedrals"].shape[0]
        nsteps = min(maxiter, max(1, int(N / self.p.batch_size)))
        refs = self._engine.reference_costs(nsteps, d["central_angles"].mean(0), d["central_dihedrals"].mean(0),
                                            d["central_distances"].mean(0))
        s = {k: float(np.mean(v)) for k, v in refs.items()}
        print(f"After {nsteps - 1} steps setting cost references: {s} to parameters.")
        self.p.angle_cost_reference = s["angle_cost"]
        self.p.dihedral_cost_reference = s["dihedral_cost"]
        self.p.cartesian_cost_reference = s["cartesian_cost"]
        if not self.read_only:
            self.p.save()
        return refs

    # -- inference ----------------------------------------------------------------------------------------
    def encode(self, data=None) -> np.ndarray:
        """data: None (train data), a list of the encoder input arrays, a dict / object with ``.CVs``, or a single
        dihedral array when only dihedrals are used (autoencoder.py:1110-1152, :2403-2465)."""
        if data is None:
            arrays = self.train_data
        elif isinstance(data, (list, tuple)):
            arrays = [np.asarray(a) for a in data]
        elif isinstance(data, np.ndarray):
            arrays = [data]
        else:
            arrays = self._encoder_arrays(_as_cv_dict(data))
        return self._engine.encode(arrays)

    def decode(self, data: np.ndarray):
        out = self._engine.decode(np.asarray(data))
        return out if len(out) > 1 else out[0]

    def generate(self, points: np.ndarray, top=None, backend=None, progbar=None) -> np.ndarray:
        """Back-maps latent points to Cartesians (autoencoder.py:2467-2576).

        ``top=None``: backbone (central) Cartesians [n, A, 3] from the decoded dihedrals (and angles when they are learned,
        the data-set mean angles otherwise) with the data-set mean bond lengths.
        ``top=AllAtomTopology``: all-atom conformations [n, A_all, 3]: the template frame with every central (and
        side-chain) torsion set to the decoded value, as ``mdtraj_backmapping`` does (misc/backmapping.py:1060-1787).
        Building MDTraj/MDAnalysis objects from topology files is out of scope; arrays are returned."""
        out = self._engine.decode(np.asarray(points, np.float32))
        if top is not None:
            from .backmapping import AllAtomTopology

            if not isinstance(top, AllAtomTopology):
                raise ValueError(f"Type of argument `top` must be AllAtomTopology (topology files need mdtraj/MDAnalysis: out of scope). "
                                 f"You provided {type(top)}.")
            if len(out) == 1:
                dihedrals, side = out[0], None
            elif len(out) == 2:
                dihedrals, side = out[1], None
            else:
                dihedrals, side = out[1], out[2]
            return top.backmap(dihedrals, side if top.n_side else None, device=str(self._engine.device))
        d = self.inp_CV_data
        if self.p.use_backbone_angles:
            angles, dihedrals = out[0], out[1]
        else:
            angles, dihedrals = d["central_angles"].mean(0, keepdims=True), out[0]
        return self._engine.backmap(angles, dihedrals, d["central_distances"].mean(0))

    @classmethod
    def from_cartesians(cls, central_cartesians: np.ndarray, parameters: Optional[ADCParameters] = None, **kw):
        """Builds the model from raw backbone coordinates [N, A, 3]: lengths, angles and dihedrals are computed on the
        device (``features.featurize_backbone``)."""
        from .features import featurize_backbone

        return cls(featurize_backbone(central_cartesians, device=kw.get("device") or "cuda:0"), parameters, **kw)

    @classmethod
    def from_checkpoint(cls, trajs, checkpoint_path: Union[str, Path], dataset=None, use_previous_model: bool = False,
                        compat: bool = False, **engine_kw):
        file = ckpt.resolve(checkpoint_path)
        parameter_file = file.parent / "parameters.json"
        if not parameter_file.is_file():
            raise Exception(f"There was no parameters.json file in the directory. {parameter_file.parent}")
        p = ADCParameters.from_file(parameter_file)
        p.cartesian_cost_scale_soft_start = tuple(p.cartesian_cost_scale_soft_start)
        p = ckpt.reconcile_step(file, p, use_previous_model)
        payload = ckpt.load_checkpoint(file)
        n_layers = sum(1 for k in payload if k.endswith("/kernel"))
        arrays = []
        for i in range(n_layers):
            stem = "dense" if i == 0 else f"dense_{i}"
            arrays += [payload[f"{stem}/kernel"], payload[f"{stem}/bias"]]
        new = cls(trajs, p, model=arrays, read_only=False, dataset=dataset, **engine_kw)
        if "adam/m" in payload and payload["adam/m"].shape == (new._engine.layout.n_params,):
            new._engine.load_state_dict({"m": payload["adam/m"], "v": payload["adam/v"], "step": int(payload["adam/iterations"])})
        return new
