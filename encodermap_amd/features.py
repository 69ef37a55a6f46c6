"""Backbone featurisation on the device ("next" row 2).

The reference turns MD coordinates into the ADC model's inputs through mdtraj's C kernels
(encodermap/loading/features.py:150-156, 1133, 1394).  Here the array math of that step -- bond lengths, bond angles
and dihedrals of consecutive backbone atoms -- is one HIP kernel, so that ``AngleDihedralCartesianEncoderMap`` can be
fed from raw ``central_cartesians [N, A, 3]`` alone.  File I/O and topology handling stay out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def featurize_backbone(central_cartesians: np.ndarray, device: str = "cuda:0", chunk: int = 1 << 18) -> dict:
    """central_cartesians [N, A, 3] (atoms in chain order) -> dict with the reference's CV names:
    ``central_cartesians``, ``central_distances [N, A-1]``, ``central_angles [N, A-2]`` (rad),
    ``central_dihedrals [N, A-3]`` (rad, mdtraj sign convention)."""
    _lib.require_gpu()
    xyz = np.ascontiguousarray(central_cartesians, dtype=np.float32)
    if xyz.ndim != 3 or xyz.shape[2] != 3 or xyz.shape[1] < 4:
        raise Exception(f"central_cartesians must have shape [N, A >= 4, 3], got {xyz.shape}")
    N, A = xyz.shape[:2]
    out = {"central_cartesians": xyz, "central_distances": np.empty((N, A - 1), np.float32),
           "central_angles": np.empty((N, A - 2), np.float32), "central_dihedrals": np.empty((N, A - 3), np.float32)}
    dev = torch.device(device)
    for s in range(0, N, chunk):
        x = torch.from_numpy(xyz[s: s + chunk]).to(dev)
        n = x.shape[0]
        d = torch.empty(n, A - 1, dtype=torch.float32, device=dev)
        a = torch.empty(n, A - 2, dtype=torch.float32, device=dev)
        t = torch.empty(n, A - 3, dtype=torch.float32, device=dev)
        _lib.backbone_features(x, d, a, t)
        out["central_distances"][s: s + chunk] = d.cpu().numpy()
        out["central_angles"][s: s + chunk] = a.cpu().numpy()
        out["central_dihedrals"][s: s + chunk] = t.cpu().numpy()
    return out
