"""All-atom back-mapping on the device ("next" row 3).

The reference rebuilds all-atom conformations from decoded torsions in ``mdtraj_backmapping``
(encodermap/misc/backmapping.py:1060-1787): every output frame starts as a copy of one template frame and has its
central dihedrals, then its side-chain dihedrals, set one after the other by rotating the atoms on the far side of each
torsion's central bond (:1668-1690, :1722-1745; numba variant ``parallel_rotation_application`` :383-405).  Which atoms
are "far" comes from cutting the bond graph at that bond (``_get_near_and_far_networkx``, misc/rotate.py:409-511).

Here the rotations are one HIP kernel (``emx_allatom_backmap``: one wave per frame, the frame in LDS); the topology is
supplied by the caller as plain arrays -- atom quadruplets of the torsions and the bond list -- since reading topologies
(mdtraj / MDAnalysis) is out of scope.  ``TorsionTable`` does the graph cut on the host once per topology.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib


def _far_side(n_atoms: int, adj: list, a: int, b: int) -> np.ndarray:
    """Boolean mask of the atoms still connected to ``b`` once the bond a-b is cut; raises like the reference when the
    bond does not exist or the cut leaves the molecule in one piece (ring) / the graph had several pieces to begin with."""
    if b not in adj[a]:
        raise Exception(f"Seems like the edge [{a}, {b}] is not part of the graph.")
    seen = np.zeros(n_atoms, bool)
    seen[b] = True
    stack = [b]
    while stack:
        u = stack.pop()
        for v in adj[u]:
            if (u == b and v == a) or (u == a and v == b) or seen[v]:
                continue
            seen[v] = True
            stack.append(v)
    if seen[a]:
        raise Exception(f"Splitting at edge [{a}, {b}] does not work: the two atoms are still connected (cyclic).")
    return seen


class TorsionTable:
    """Torsions of one topology in the layout the kernel reads.

    n_atoms     number of atoms A
    bonds       int [n_bonds, 2]   the bond graph (must be connected and the torsion bonds must not lie in rings)
    quadruplets int [J, 4]         atoms of each torsion, in the order the reference applies them (central dihedrals first,
                                   then side chains); the rotated bond is quadruplets[:, 1:3] (backmapping.py:1509)
    active      optional bool [J]  torsions to skip (False) -- the reference drops proline N-CA / CA-CB / CB-CG this way
                                   (:1526-1538, :1570-1597); the matching columns of the targets are ignored
    """

    def __init__(self, n_atoms: int, bonds, quadruplets, active: Optional[Sequence[bool]] = None):
        self.n_atoms = int(n_atoms)
        bonds = np.asarray(bonds, np.int64).reshape(-1, 2)
        quads = np.asarray(quadruplets, np.int64).reshape(-1, 4)
        if quads.size and (quads.min() < 0 or quads.max() >= self.n_atoms):
            raise ValueError("torsion atom index out of range")
        if bonds.size and (bonds.min() < 0 or bonds.max() >= self.n_atoms):
            raise ValueError("bond atom index out of range")
        self.active = np.ones(len(quads), bool) if active is None else np.asarray(active, bool)
        if self.active.shape != (len(quads),):
            raise ValueError(f"`active` must have one entry per torsion ({len(quads)}), got shape {self.active.shape}")
        adj = [set() for _ in range(self.n_atoms)]
        for u, v in bonds:
            adj[int(u)].add(int(v))
            adj[int(v)].add(int(u))
        # the whole molecule must be one piece (backmapping.py:1486-1506 raises otherwise)
        seen = np.zeros(self.n_atoms, bool)
        stack = [0]
        seen[0] = True
        while stack:
            u = stack.pop()
            for v in adj[u]:
                if not seen[v]:
                    seen[v] = True
                    stack.append(v)
        if not seen.all():
            raise Exception(f"the bond graph has more than one connected component ({int((~seen).sum())} atoms unreachable from atom 0)")
        self.quadruplets = quads[self.active]
        self.far_sides = [np.flatnonzero(_far_side(self.n_atoms, adj, int(q[1]), int(q[2]))) for q in self.quadruplets]
        self._dev = {}

    @property
    def n_torsions(self) -> int:
        return len(self.quadruplets)

    def packed_far_mask(self) -> np.ndarray:
        """uint32 [J, W, 64]: bit k of [j, w, l] <-> atom (32 w + k) 64 + l on the far side of torsion j (include/emx.h)."""
        W = (self.n_atoms + 2047) // 2048
        m = np.zeros((self.n_torsions, W, 64), np.uint32)
        for j, far in enumerate(self.far_sides):
            blk, lane = far // 64, far % 64
            np.bitwise_or.at(m[j], (blk // 32, lane), (np.uint32(1) << (blk % 32).astype(np.uint32)))
        return m

    def device_tables(self, device):
        key = str(device)
        if key not in self._dev:
            q = torch.from_numpy(np.ascontiguousarray(self.quadruplets, dtype=np.int32)).to(device)
            m = torch.from_numpy(self.packed_far_mask().view(np.int32)).to(device)
            self._dev[key] = (q, m)
        return self._dev[key]


def allatom_backmap(template_xyz: np.ndarray, table: TorsionTable, dihedrals: np.ndarray, device: str = "cuda:0",
                    chunk: int = 1 << 16) -> np.ndarray:
    """template_xyz [A, 3] (one frame; every output frame starts from it) or [n, A, 3]; dihedrals [n, J_all] in rad, one
    column per entry of the table's ``active`` mask -> float32 [n, A, 3]."""
    _lib.require_gpu()
    tmpl = np.ascontiguousarray(template_xyz, dtype=np.float32)
    tg = np.asarray(dihedrals, np.float32)
    if tg.ndim != 2 or tg.shape[1] != len(table.active):
        raise Exception(f"dihedrals must have shape [n, {len(table.active)}], got {tg.shape}")
    if tmpl.shape[-2:] != (table.n_atoms, 3) or tmpl.ndim not in (2, 3) or (tmpl.ndim == 3 and tmpl.shape[0] != tg.shape[0]):
        raise Exception(f"template_xyz must have shape [{table.n_atoms}, 3] or [{tg.shape[0]}, {table.n_atoms}, 3], got {tmpl.shape}")
    tg = np.ascontiguousarray(tg[:, table.active])
    n, A = tg.shape[0], table.n_atoms
    dev = torch.device(device)
    quads, mask = table.device_tables(dev)
    out = np.empty((n, A, 3), np.float32)
    t_dev = torch.from_numpy(tmpl).to(dev) if tmpl.ndim == 2 else None
    for s in range(0, n, chunk):
        t = torch.from_numpy(tg[s: s + chunk]).to(dev)
        x0 = t_dev if t_dev is not None else torch.from_numpy(tmpl[s: s + chunk]).to(dev)
        o = torch.empty(t.shape[0], A, 3, dtype=torch.float32, device=dev)
        _lib.allatom_backmap(x0, quads, mask, t, o)
        out[s: s + chunk] = o.cpu().numpy()
    return out


class AllAtomTopology:
    """What ``generate(points, top=...)`` needs of a topology, as arrays (the reference reads it from mdtraj objects,
    autoencoder.py:2535-2556): one template frame, the bond graph, and the atom quadruplets of the model's outputs --
    ``central_quadruplets[j]`` belongs to column j of the decoded central dihedrals, ``side_quadruplets[j]`` to column j of
    the decoded side-chain dihedrals."""

    def __init__(self, template_xyz, bonds, central_quadruplets, side_quadruplets=None, central_active=None, side_active=None):
        self.template_xyz = np.ascontiguousarray(template_xyz, dtype=np.float32)
        if self.template_xyz.ndim != 2 or self.template_xyz.shape[1] != 3:
            raise Exception(f"template_xyz must have shape [A, 3], got {self.template_xyz.shape}")
        cq = np.asarray(central_quadruplets, np.int64).reshape(-1, 4)
        sq = np.zeros((0, 4), np.int64) if side_quadruplets is None else np.asarray(side_quadruplets, np.int64).reshape(-1, 4)
        ca = np.ones(len(cq), bool) if central_active is None else np.asarray(central_active, bool)
        sa = np.ones(len(sq), bool) if side_active is None else np.asarray(side_active, bool)
        self.n_central, self.n_side = len(cq), len(sq)
        self.table = TorsionTable(len(self.template_xyz), bonds, np.concatenate([cq, sq]), np.concatenate([ca, sa]))

    def backmap(self, dihedrals: np.ndarray, sidechain_dihedrals: Optional[np.ndarray] = None, device: str = "cuda:0") -> np.ndarray:
        d = np.asarray(dihedrals, np.float32)
        if d.ndim != 2 or d.shape[1] != self.n_central:
            raise Exception(f"dihedrals must have shape [n, {self.n_central}], got {d.shape}")
        if self.n_side:
            if sidechain_dihedrals is None:
                raise Exception("this topology lists side-chain torsions but no sidechain_dihedrals were given")
            s = np.asarray(sidechain_dihedrals, np.float32)
            if s.shape != (d.shape[0], self.n_side):
                raise Exception(f"sidechain_dihedrals must have shape [{d.shape[0]}, {self.n_side}], got {s.shape}")
            d = np.concatenate([d, s], axis=1)
        return allatom_backmap(self.template_xyz, self.table, d, device=device)
