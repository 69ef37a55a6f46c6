"""Checkpoint files: ``saved_model_{step}.emx.npz`` next to ``parameters.json``.

Same cadence, naming stem and resume rules as the reference's ``save_model``/``load_model``
(encodermap/misc/saving_loading_models.py:201-268, 360-626): newest file by trailing number when a directory
is given, refusal to load an intermediate step unless ``use_previous_model=True`` (which backs up
``parameters.json`` and the later checkpoints).  The container is NumPy's npz (a Keras ``.keras`` archive cannot
be produced without Keras): kernels/biases under the reference's initializer-dict keys (``dense/kernel``,
``dense/bias``, ``dense_1/kernel`` ..., models.py:113-248) plus the Adam moments and the step, which the
reference does not restore (it builds a fresh Adam, autoencoder.py:741).
"""
from __future__ import annotations

import datetime
import re
import shutil
from pathlib import Path
from typing import Optional

import numpy as np

SUFFIX = ".emx.npz"


def _now() -> str:
    return datetime.datetime.now().astimezone().replace(microsecond=0).isoformat().replace(":", "-")


def checkpoint_name(main_path, step: Optional[int]) -> Path:
    stem = f"saved_model_{step}" if step is not None else f"saved_model_{_now()}"
    return Path(main_path) / (stem + SUFFIX)


def step_of(path: Path) -> Optional[int]:
    m = re.fullmatch(r"saved_model_(\d+)" + re.escape(SUFFIX), Path(path).name)
    return int(m.group(1)) if m else None


def find_checkpoints(directory) -> list:
    """Checkpoint files of a run directory, oldest first (numbered ones by step, time-stamped ones last by mtime)."""
    files = list(Path(directory).glob("saved_model_*" + SUFFIX))
    numbered = sorted((f for f in files if step_of(f) is not None), key=step_of)
    stamped = sorted((f for f in files if step_of(f) is None and "_backup_from_" not in f.name), key=lambda f: f.stat().st_mtime)
    return numbered + stamped


def save_checkpoint(path: Path, layer_arrays, state: dict, class_name: str) -> Path:
    payload = {}
    for i in range(len(layer_arrays) // 2):
        stem = "dense" if i == 0 else f"dense_{i}"
        payload[f"{stem}/kernel"] = layer_arrays[2 * i]
        payload[f"{stem}/bias"] = layer_arrays[2 * i + 1]
    payload["adam/m"] = state["m"]
    payload["adam/v"] = state["v"]
    payload["adam/iterations"] = np.int64(state["step"])
    payload["class_name"] = np.array(class_name)
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        np.savez(f, **payload)
    return path


def load_checkpoint(path: Path) -> dict:
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def resolve(checkpoint_path) -> Path:
    p = Path(checkpoint_path)
    if p.is_dir():
        found = find_checkpoints(p)
        if not found:
            raise Exception(f"checkpoint_path={p} has no saved_model_*{SUFFIX} files")
        return found[-1]
    if not p.is_file():
        raise Exception(f"{p} is neither a directory with checkpoints nor a checkpoint file")
    return p


def reconcile_step(ckpt: Path, p, use_previous_model: bool):
    """The intermediate-step rule of the reference: the ADC (``cartesian``) branch resumes at ``step + 1``
    (saving_loading_models.py:542-559), the plain Autoencoder / EncoderMap branch at ``step`` (597-616)."""
    step = step_of(ckpt)
    if step is None or step == p.current_training_step:
        return p
    parameter_file = ckpt.parent / "parameters.json"
    backup = ckpt.parent / f"parameters_at_{p.current_training_step}_{_now()}_.json"
    if not use_previous_model:
        raise Exception(
            f"The model was saved at step {step}, but the parameters file has its current step at "
            f"{p.current_training_step}. It seems like you are reloading a model at an intermediate step. If you set "
            f"the `use_previous_model` flag to True, I will backup the parameters file to {backup} and set the new "
            f"training step so, that you can use/retrain this model.")
    if parameter_file.is_file():
        shutil.move(parameter_file, backup)
    for f in find_checkpoints(ckpt.parent):
        s = step_of(f)
        if s is not None and s > step:
            shutil.copyfile(f, f.parent / (f.name[: -len(SUFFIX)] + f"_backup_from_{step}_{_now()}" + SUFFIX))
    resume = step + 1 if hasattr(p, "cartesian_cost_scale") else step
    p.current_training_step = resume
    p.n_steps = resume
    p.save()
    return p
