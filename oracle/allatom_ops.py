"""TEST INFRASTRUCTURE -- CPU restatement of the reference's all-atom back-mapping arithmetic.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this package; the
product (``encodermap_amd``) never does.

What is restated, in NumPy, following the reference line by line in meaning (not in text):

* ``dihedral``            encodermap/misc/rotate.py:547-581 (the mdtraj formula the reference adapted)
* ``rotation_matrix``     ``transformations.rotation_matrix(angle, direction, point)`` -- third-party, un-vendored
                          (``transformations`` by C. Gohlke, unpinned in the reference's md_requirements); its published
                          algorithm: R = cos a I + (1 - cos a) d d^T + sin a [d]_x with d the unit direction, and the
                          translation column  point - R point.  Known answers from that package's docstring are pinned in
                          tests/test_oracle_kat.py.
* ``far_side``            encodermap/misc/rotate.py:409-511: remove the bond from the bond graph; the far side is the connected
                          component that holds the bond's SECOND atom (exactly two components are required)
* ``allatom_backmap``     encodermap/misc/backmapping.py:1618-1627 (state: float32 coordinates padded with ones) and the
                          serial loops :1668-1690 (central dihedrals) and :1722-1745 (side-chain dihedrals): for every
                          frame, dihedral after dihedral: current angle, rotate the far side about the central bond by
                          target - current (float64 matrix applied to the float32 state, stored back as float32).

parity: the reference's own tests of this routine need mdtraj and downloaded trajectories, so no golden vector of the
complete routine exists here -- "parity unpinned" beyond the pieces named above; the GPU tests add the size-independent
property the reference itself asserts with ``verify_every_rotation`` (every requested dihedral is realised to 1e-3 rad).
"""
from __future__ import annotations

import numpy as np


def dihedral(xyz: np.ndarray, quad) -> float:
    """Dihedral (rad, in (-pi, pi]) of atoms quad[0..3] of one frame xyz [A, 3]; rotate.py:547-581."""
    p = np.asarray(xyz)[np.asarray(quad)]
    b1, b2, b3 = p[1] - p[0], p[2] - p[1], p[3] - p[2]
    c1, c2 = np.cross(b2, b3), np.cross(b1, b2)
    p1 = (b1 * c1).sum(-1)
    p1 = p1 * (b2 * b2).sum(-1) ** 0.5
    p2 = (c1 * c2).sum(-1)
    return np.arctan2(p1, p2)


def rotation_matrix(angle: float, direction, point=None) -> np.ndarray:
    """4x4 homogeneous matrix rotating by ``angle`` about the axis through ``point`` along ``direction``."""
    sina, cosa = np.sin(np.float64(angle)), np.cos(np.float64(angle))
    d = np.array(direction[:3], dtype=np.float64)
    d = d / np.sqrt(np.dot(d, d))
    R = np.diag([cosa, cosa, cosa])
    R += np.outer(d, d) * (1.0 - cosa)
    d = d * sina
    R += np.array([[0.0, -d[2], d[1]], [d[2], 0.0, -d[0]], [-d[1], d[0], 0.0]])
    M = np.identity(4)
    M[:3, :3] = R
    if point is not None:
        pt = np.array(point[:3], dtype=np.float64)
        M[:3, 3] = pt - np.dot(R, pt)
    return M


def far_side(n_atoms: int, bonds: np.ndarray, edge) -> np.ndarray:
    """Atoms on the far side of ``edge`` = the connected component of ``edge[1]`` once the edge is removed from the bond
    graph; raises if the edge is no bond or if removing it does not split the molecule in two (rotate.py:443-497)."""
    a, b = int(edge[0]), int(edge[1])
    adj = [[] for _ in range(n_atoms)]
    found = False
    for u, v in np.asarray(bonds):
        u, v = int(u), int(v)
        if {u, v} == {a, b}:
            found = True
            continue
        adj[u].append(v)
        adj[v].append(u)
    if not found:
        raise Exception(f"the edge {a}-{b} is not part of the bond graph")
    comp = np.full(n_atoms, -1)
    n_comp = 0
    for s in range(n_atoms):
        if comp[s] >= 0:
            continue
        stack = [s]
        comp[s] = n_comp
        while stack:
            u = stack.pop()
            for v in adj[u]:
                if comp[v] < 0:
                    comp[v] = n_comp
                    stack.append(v)
        n_comp += 1
    if n_comp != 2 or comp[a] == comp[b]:
        raise Exception(f"splitting at edge {a}-{b} gives {n_comp} components (cyclic, or more than one chain)")
    return np.flatnonzero(comp == comp[b])


def allatom_backmap(template: np.ndarray, quads: np.ndarray, far_sides: list, dihedrals: np.ndarray,
                    side_quads: np.ndarray = None, side_far_sides: list = None, side_dihedrals: np.ndarray = None) -> np.ndarray:
    """template [A, 3] (one frame, copied for every output frame as the reference joins copies of frame 0,
    backmapping.py:1607-1613); quads [J, 4]; far_sides: J index arrays; dihedrals [n, J] -> xyz [n, A, 3] float32."""
    dihedrals = np.asarray(dihedrals, np.float32)
    n = dihedrals.shape[0]
    A = template.shape[0]
    xyz = np.ones((n, A, 4), np.float32)
    xyz[:, :, :3] = np.asarray(template, np.float32)[None]
    groups = [(np.asarray(quads), far_sides, dihedrals)]
    if side_dihedrals is not None:
        groups.append((np.asarray(side_quads), side_far_sides, np.asarray(side_dihedrals, np.float32)))
    for i in range(n):
        for q, far, tgt in groups:
            for j in range(tgt.shape[1]):
                quad = q[j]
                bond = quad[1:3]
                current = dihedral(xyz[i, :, :3], quad)
                angle = tgt[i, j] - current
                direction = xyz[i, bond[1], :3] - xyz[i, bond[0], :3]
                pivot = xyz[i, bond[0], :3]
                M = rotation_matrix(angle, direction, pivot)
                xyz[i, far[j]] = M.dot(xyz[i, far[j]].T).T          # float64 product stored into the float32 state
    return np.ascontiguousarray(xyz[..., :3])
