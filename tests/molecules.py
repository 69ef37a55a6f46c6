"""Synthetic tree-shaped molecules for the all-atom back-mapping tests (no reference data involved)."""
import numpy as np


def random_molecule(n_backbone: int, seed: int, side_every: int = 3, max_side: int = 4, leaves: bool = True):
    """A backbone chain with side chains and leaf atoms ("hydrogens"), atoms numbered in a shuffled order so that far sides
    are no index ranges.  Returns (xyz [A, 3] float32, bonds [n_bonds, 2], central_quads [Jc, 4], side_quads [Js, 4])."""
    rng = np.random.default_rng(seed)
    parent = [-1]
    backbone = [0]
    for _ in range(1, n_backbone):
        parent.append(backbone[-1])
        backbone.append(len(parent) - 1)
    side_quads = []
    for bi in range(1, n_backbone - 1):
        if bi % side_every:
            continue
        chain = [backbone[bi - 1], backbone[bi]]
        for _ in range(int(rng.integers(1, max_side + 1))):
            parent.append(chain[-1])
            chain.append(len(parent) - 1)
        for k in range(len(chain) - 3):
            side_quads.append(chain[k: k + 4])
    if leaves:
        for a in range(len(parent)):
            if rng.random() < 0.5:
                parent.append(a)
    A = len(parent)
    xyz = np.zeros((A, 3))
    for a in range(1, A):
        while True:
            step = rng.normal(size=3)
            step *= 0.15 / np.linalg.norm(step)
            cand = xyz[parent[a]] + step
            # keep clear of the other atoms so that no torsion's bonds are (nearly) collinear
            if a < 2 or np.min(np.linalg.norm(xyz[:a] - cand, axis=1)) > 0.08:
                break
        xyz[a] = cand
    perm = rng.permutation(A)                      # new index of old atom a = perm[a]
    xyz_p = np.zeros_like(xyz)
    xyz_p[perm] = xyz
    bonds = np.array([[perm[parent[a]], perm[a]] for a in range(1, A)])
    central = np.array([[perm[backbone[i + k]] for k in range(4)] for i in range(n_backbone - 3)])
    side = np.array([[perm[a] for a in q] for q in side_quads]).reshape(-1, 4)
    return xyz_p.astype(np.float32), bonds, central, side


def dihedrals_of(xyz: np.ndarray, quads: np.ndarray) -> np.ndarray:
    """xyz [n, A, 3] float64 math -> [n, J] (same formula as the oracle, vectorised)."""
    p = np.asarray(xyz, np.float64)[:, quads]          # [n, J, 4, 3]
    b1, b2, b3 = p[:, :, 1] - p[:, :, 0], p[:, :, 2] - p[:, :, 1], p[:, :, 3] - p[:, :, 2]
    c1, c2 = np.cross(b2, b3), np.cross(b1, b2)
    return np.arctan2((b1 * c1).sum(-1) * np.sqrt((b2 * b2).sum(-1)), (c1 * c2).sum(-1))
