"""Synthetic train data of the reference's API surface (host-side NumPy, no GPU).

``create_n_cube``        encodermap/misc/misc.py:596-695   default train data when ``train_data=None``
``random_on_cube_edges`` encodermap/encodermap_tf1/misc.py:246-283   the cube example (examples/cube.py)
"""
from __future__ import annotations

import itertools
from typing import Optional

import numpy as np


def _hypercube_edges(n: int) -> np.ndarray:
    """Edges of the n-cube in the node/edge order of ``networkx.hypercube_graph(n)``: nodes are the n-tuples
    over {0,1} in lexicographic order, each node lists its neighbours dimension by dimension, an edge is
    reported once, from the endpoint that is visited first."""
    nodes = list(itertools.product((0, 1), repeat=n))
    seen, edges = set(), []
    for a in nodes:
        for d in range(n):
            b = list(a)
            b[d] = 1 - b[d]
            b = tuple(b)
            if b not in seen:
                edges.append((a, b))
        seen.add(a)
    return np.array(edges)


def create_n_cube(n: int = 3, points_along_edge: int = 500, sigma: float = 0.05, same_colored_edges: int = 3,
                  seed: Optional[int] = None) -> tuple:
    """Points along the edges of the n-dimensional unit hypercube with Gaussian noise ``sigma``; returns
    (coordinates [E*points_along_edge, n], colour ids).  n=4 gives the (16000, 4) default data set."""
    if seed is not None:
        np.random.seed(seed=seed)
    edges = _hypercube_edges(n)
    lin = np.linspace(0, 1, points_along_edge)
    coords, ids = [], []
    for i, (A, B) in enumerate(edges):
        pts = A + ((B - A)[:, None] * lin).T
        if sigma:
            pts = pts + np.random.normal(scale=sigma, size=(len(pts), n))
        coords.append(pts)
        ids.append(np.full(len(pts), float(i)))
    coords = np.concatenate(coords)
    ids = np.concatenate(ids)
    # colour `same_colored_edges` pairs of edges that share their first vertex alike
    found, pairs = [], []
    for _ in range(same_colored_edges):
        for i, edge in enumerate(edges):
            if i in found:
                continue
            found.append(i)
            j = i
            for j in np.where(np.all(edges[:, 0] == edge[0], axis=1))[0]:
                if j not in found:
                    found.append(j)
                    break
            if i != j:
                pairs.append((i, j))
                break
    for i, j in pairs:
        ids[ids == i] = j
    return coords, ids


def random_on_cube_edges(n_points: int, sigma: float = 0, rng: Optional[np.random.Generator] = None):
    """Uniform points on the 12 edges of the unit cube (+ Gaussian noise); returns (coordinates [n,3], edge ids)."""
    uniform = rng.uniform if rng is not None else np.random.uniform
    normal = rng.normal if rng is not None else np.random.normal
    r = uniform(size=n_points)
    starts = np.array([[0, 0, 0]] * 3 + [[1, 1, 0]] * 3 + [[0, 1, 1]] * 3 + [[1, 0, 1]] * 3, float)
    dirs = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, 1],
                     [1, 0, 0], [0, -1, 0], [0, 0, -1], [-1, 0, 0], [0, 1, 0], [0, 0, -1]], float)
    coords = np.zeros((n_points, 3))
    ids = np.zeros(n_points)
    for i in range(12):
        mask = (i / 12 < r) & (r < (i + 1) / 12)
        coords[mask] += starts[i] + (r[mask, None] - i / 12) * 12 * dirs[i]
        ids[mask] = i
    if sigma:
        coords += normal(scale=sigma, size=(n_points, 3))
    return coords, ids
