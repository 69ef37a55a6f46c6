"""``Parameters`` / ``ADCParameters`` -- the configuration objects of the drop-in API.

Mirrors the interface of the reference's ``encodermap/parameters/parameters.py`` (class framework :154-503,
``Parameters._defaults`` :611-639, ``ADCParameters._defaults`` :794-828): identical attribute names, default
values and JSON keys, attribute *and* item access, ``save``/``load``/``from_file``/``update``/``to_dict``,
unknown keys dropped with a message, the ``n_neurons``/``activation_functions`` length check and the legacy
``n_epochs`` migration.  Engine-only knobs (``EngineOptions``) live outside these classes so that a
``parameters.json`` written here loads in the reference and vice versa.
"""
from __future__ import annotations

import copy
import datetime
import json
import os
from math import pi
from pathlib import Path
from textwrap import wrap
from typing import Any, Optional, Union

__all__ = ["Parameters", "ADCParameters"]


def _timestamp() -> str:
    return datetime.datetime.now().astimezone().replace(microsecond=0).isoformat().replace(":", "-")


def _table(rows: list[dict[str, Any]]) -> str:
    if not rows:
        return ""
    cols = list(rows[0].keys())
    cells = [[str(r[c]).split("\n") for c in cols] for r in rows]
    widths = [max(len(c), *(len(line) for row in cells for line in row[i])) for i, c in enumerate(cols)]
    sep = "+" + "+".join("-" * (w + 2) for w in widths) + "+"
    out = [sep, "| " + " | ".join(c.ljust(w) for c, w in zip(cols, widths)) + " |", sep.replace("-", "=")]
    for row in cells:
        for k in range(max(len(c) for c in row)):
            out.append("| " + " | ".join((c[k] if k < len(c) else "").ljust(w) for c, w in zip(row, widths)) + " |")
        out.append(sep)
    return "\n".join(out)


_DESCRIPTIONS = {
    "n_neurons": "Neurons per encoder layer; the last entry is the bottleneck. The decoder mirrors it.",
    "activation_functions": "Activation per layer, len(n_neurons) + 1 entries; '' means linear. First entry is the input layer.",
    "periodicity": "Period of the input space (2*pi for radians, 360 for degrees, float('inf') for none).",
    "learning_rate": "Adam learning rate.",
    "n_steps": "Number of training steps.",
    "batch_size": "Points per training step.",
    "summary_step": "A summary (progress, loss) is produced every summary_step steps.",
    "checkpoint_step": "A checkpoint is written every checkpoint_step steps.",
    "dist_sig_parameters": "(sig_h, a_h, b_h, sig_l, a_l, b_l) of the sketch-map sigmoids.",
    "distance_cost_scale": "Weight of the sketch-map distance cost; None disables it.",
    "auto_cost_scale": "Weight of the auto-encoding cost; None disables it.",
    "auto_cost_variant": "mean_square, mean_abs or mean_norm.",
    "center_cost_scale": "Weight of the center cost; None disables it.",
    "l2_reg_constant": "L2 regularisation constant on the kernels.",
    "gpu_memory_fraction": "Kept for compatibility; unused.",
    "analysis_path": "Path for analysis output.",
    "id": "Identifier of the run.",
    "model_api": "sequential or functional.",
    "loss": "emap_cost, reconstruction_loss or mse.",
    "training": "auto (custom is not implemented, as in the reference).",
    "batched": "Kept for compatibility.",
    "tensorboard": "Kept for compatibility; TensorBoard output is out of scope.",
    "seed": "Seed for weight initialisation and batch order.",
    "current_training_step": "Number of steps already trained (resume bookkeeping).",
    "write_summary": "Whether summaries are written.",
    "trainable_dense_to_sparse": "Kept for compatibility (sparse input is out of scope).",
    "using_hypercube": "True when the default hypercube train data is in use.",
}


class ParametersFramework:
    """Attribute/dict hybrid with JSON/YAML persistence (reference: parameters.py:154-503)."""

    _defaults: dict[str, Any] = {}

    def __init__(self, **kwargs: Any) -> None:
        self.main_path = os.getcwd()
        for key, value in kwargs.items():
            if key in self.defaults:
                setattr(self, key, value)
            elif key == "n_epochs":
                raise Exception("Parameter `n_epochs` and `n_steps_per_epoch` is deprecated. Use `n_steps` instead.")
            elif key == "main_path":
                setattr(self, key, value)
            else:
                print(f"Dropping unknown dict entry for {{'{key}': {value}}}")
        if len(kwargs["n_neurons"]) != len(kwargs["activation_functions"]) - 1:
            raise Exception(
                f"Length of `n_neurons` and `activation_functions` (-1) does not match: {self.n_neurons}, {self.activation_functions}"
            )

    # -- access ----------------------------------------------------------------------------
    @property
    def defaults(self) -> dict[str, Any]:
        return copy.deepcopy(self._defaults)

    def __getitem__(self, item: str) -> Any:
        return getattr(self, item)

    def __setitem__(self, key: str, value: Any) -> None:
        if key == "cartesian_cost_scale_soft_start" and value is not None and len(value) != 2:
            raise Exception("Parameter cartesian_cost_scale_soft_start only takes a tuple of len 2.")
        setattr(self, key, value)

    def update(self, **kwargs: Any) -> None:
        for key, value in kwargs.items():
            if key not in self.__dict__:
                print(f"Dropping unknown dict entry for {{'{key}': {value}}}")
            else:
                setattr(self, key, value)

    def to_dict(self) -> dict[str, Any]:
        return {k: getattr(self, k) for k in self.defaults} | {"main_path": self.main_path}

    # -- persistence -----------------------------------------------------------------------
    def save(self, path: Optional[Union[str, Path]] = None) -> str:
        path = os.path.join(self.main_path, "parameters.json") if path is None else str(path)
        fmt = path.split(".")[-1]
        if fmt not in ("json", "yaml"):
            raise OSError(f"Unrecognized extension .{fmt} in path {path}. Please provide either '.json' or '.yaml'")
        if os.path.isfile(path):
            stem, ext = os.path.splitext(path)
            os.rename(path, stem + "_back_" + _timestamp() + ext)
        with open(path, "w") as f:
            if fmt == "json":
                json.dump(self.__dict__, f, indent=4, sort_keys=True)
            else:
                import yaml

                yaml.dump(self.__dict__, f, default_flow_style=True)
        return path

    @staticmethod
    def _read(path: Path) -> dict[str, Any]:
        with open(path) as f:
            if path.suffix == ".json":
                return json.load(f)
            if path.suffix == ".yaml":
                import yaml

                return yaml.load(f, Loader=yaml.FullLoader)
        raise ValueError(f"The extension of the provided file should be `.json`, or `.yaml`. You provided {path.suffix}")

    @staticmethod
    def _migrate_epochs(params: dict[str, Any], scale_steps: bool) -> None:
        if "n_epochs" not in params:
            return
        print("Detected old definition `n_epochs` and `n_steps_per_epoch`. I will change that to `n_steps` = `n_epochs` * `n_steps_per_epoch`.")
        per = params["n_steps_per_epoch"]
        params["n_steps"] = params["n_epochs"] * per
        if scale_steps:
            ss = params.get("cartesian_cost_scale_soft_start")
            if ss is not None and tuple(ss) != (None, None):
                params["cartesian_cost_scale_soft_start"] = (ss[0] * per, ss[1] * per)
            params["summary_step"] *= per
            params["checkpoint_step"] *= per
        del params["n_epochs"], params["n_steps_per_epoch"]

    @classmethod
    def from_dict(cls, params: dict[str, Any]):
        return cls(**params)

    @classmethod
    def from_file(cls, path: Union[str, Path]):
        """Read a parameters file; a file that was moved gets its ``main_path`` rewritten and is re-saved."""
        path = Path(path)
        params = cls._read(path)
        cls._migrate_epochs(params, scale_steps=True)
        if Path(params["main_path"]).resolve() != path.parent.resolve():
            print("Seems like the parameter file was moved to another directory. Parameter file is updated ...")
            params["main_path"] = str(path.parent)
        new = cls(**params)
        new.save()
        return new

    @classmethod
    def load(cls, path: Union[str, Path]):
        path = Path(path)
        params = cls._read(path)
        cls._migrate_epochs(params, scale_steps=False)
        if params["main_path"] != str(path.parent):
            print("Seems like the parameter file was moved to another directory. Parameter file is updated ...")
            params["main_path"] = str(path.parent)
        return cls(**params)

    # -- presentation ----------------------------------------------------------------------
    @classmethod
    def _descr(cls, key: str) -> str:
        return "\n".join(wrap(_DESCRIPTIONS.get(key, "See the reference documentation of this ADC parameter."), width=50))

    @property
    def parameters(self) -> str:
        return _table([{"Parameter": k, "Value": getattr(self, k), "Description": self._descr(k)} for k in self.defaults])

    @classmethod
    def defaults_description(cls) -> str:
        return _table([{"Parameter": k, "Default Value": v, "Description": cls._descr(k)} for k, v in cls._defaults.items()])

    def _string_summary(self) -> str:
        d = self.defaults
        diff = [k for k in d if self.__dict__.get(k) != d[k]]
        s = f"{self.__class__.__name__} class with 'main_path' at {self.main_path}."
        for k in diff:
            s += f"\nNon-standard value of {k}: {self.__dict__[k]} (standard is {d[k]})"
        if not diff:
            s += " All parameters are set to default values."
        return s

    def __str__(self) -> str:
        return self._string_summary()

    def __repr__(self) -> str:
        return f"<{self._string_summary()} Object at 0x{id(self):02x}>"


class Parameters(ParametersFramework):
    """Parameters of ``Autoencoder`` / ``EncoderMap`` (defaults: reference parameters.py:611-639)."""

    _defaults = dict(
        n_neurons=[128, 128, 2],
        activation_functions=["", "tanh", "tanh", ""],
        periodicity=2 * pi,
        learning_rate=0.001,
        n_steps=1000,
        batch_size=256,
        summary_step=10,
        checkpoint_step=5000,
        dist_sig_parameters=(4.5, 12, 6, 1, 2, 6),
        distance_cost_scale=500,
        auto_cost_scale=1,
        auto_cost_variant="mean_abs",
        center_cost_scale=0.0001,
        l2_reg_constant=0.001,
        gpu_memory_fraction=0,
        analysis_path="",
        id="",
        model_api="sequential",
        loss="emap_cost",
        training="auto",
        batched=True,
        tensorboard=False,
        seed=None,
        current_training_step=0,
        write_summary=False,
        trainable_dense_to_sparse=False,
        using_hypercube=False,
    )

    def __init__(self, **kwargs: Any) -> None:
        kwargs.pop("defaults", None)
        super().__init__(**{**self.defaults, **kwargs})


class ADCParameters(ParametersFramework):
    """Parameters of ``AngleDihedralCartesianEncoderMap`` (defaults: reference parameters.py:794-828)."""

    _defaults = dict(
        Parameters._defaults,
        **dict(
            track_clashes=False,
            track_RMSD=False,
            model_api="functional",
            cartesian_pwd_start=None,
            cartesian_pwd_stop=None,
            cartesian_pwd_step=None,
            use_backbone_angles=False,
            use_sidechains=False,
            angle_cost_scale=0,
            angle_cost_variant="mean_abs",
            angle_cost_reference=1,
            dihedral_cost_scale=1,
            dihedral_cost_variant="mean_abs",
            dihedral_cost_reference=1,
            side_dihedral_cost_scale=0.5,
            side_dihedral_cost_variant="mean_abs",
            side_dihedral_cost_reference=1,
            cartesian_cost_scale=1,
            cartesian_cost_scale_soft_start=(None, None),
            cartesian_cost_variant="mean_abs",
            cartesian_cost_reference=1,
            cartesian_dist_sig_parameters=Parameters._defaults["dist_sig_parameters"],
            cartesian_distance_cost_scale=1,
            auto_cost_scale=None,
            distance_cost_scale=None,
            multimer_training=None,
            multimer_topology_classes=None,
            multimer_connection_bridges=None,
            multimer_lengths=None,
            reconstruct_sidechains=False,
        ),
    )

    def __init__(self, **kwargs: Any) -> None:
        ss = kwargs.get("cartesian_cost_scale_soft_start", (None, None))
        if ss is not None and len(ss) != 2:
            raise Exception("Parameter cartesian_cost_scale_soft_start only takes a tuple of len 2.")
        kwargs.pop("defaults", None)
        super().__init__(**{**self.defaults, **kwargs})
