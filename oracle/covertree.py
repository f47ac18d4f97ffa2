"""Oracle: the cover-tree clustering of `cggp/covertree.py:26-158` (next row F3).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  A numpy restatement that keeps the reference's
sequential semantics -- the order in which points seed children, the Lloyd re-centring with its
fall-back, the r-neighbour bookkeeping with `neighbor_factor`, the per-level Voronoi
reassignment -- but stores row indices into the data instead of copies of the rows.

The reference module is numpy arithmetic only (its TensorFlow / matplotlib imports are unused by
the algorithm), yet it cannot be imported here because those imports fail; so this restatement is
pinned by hand-checkable cases and invariants in `tests/test_covertree.py`, not by reference
outputs: **parity unpinned**.
"""

import math

import numpy as np


def _dist(p, q):
    """`covertree.py:41-44`: Euclidean norm over the last axis (the distance argument is ignored)."""
    return np.linalg.norm(p - q, axis=-1)


class Node:
    def __init__(self, point, radius, parent, rows):
        self.point = point
        self.radius = radius
        self.parent = parent
        self.rows = rows  # indices of the data rows the node holds, in the reference's order
        self.children = []
        self.r_neighbors = [self]  # :20-22
        self.voronoi_rows = None


class CoverTree:
    def __init__(self, data, spatial_resolution=None, num_levels=1, lloyds=True, voronoi=True):
        x, y = data
        self.x, self.y = x, y
        n = x.shape[0]
        centre = x.mean(axis=-2)  # :50
        max_radius = np.max(_dist(centre, x))  # :51-52
        if spatial_resolution is not None:  # :54-56
            num_levels = math.ceil(math.log2(max_radius / spatial_resolution)) + 1
            max_radius = spatial_resolution * (2 ** (num_levels - 1))
        if num_levels < 1:
            raise ValueError("cover tree needs at least one level")
        root = Node(centre, max_radius, None, np.arange(n))
        if voronoi:
            root.voronoi_rows = np.arange(n)  # :59-60
        self.levels = [[] for _ in range(num_levels)]
        self.levels[0].append(root)
        factor = 4 * (1 - 1 / 2 ** np.arange(num_levels, -1, -1))  # :65

        for level in range(1, num_levels):
            radius = max_radius / (2 ** level)  # :68
            for parent in self.levels[level - 1]:
                self._spawn_children(parent, radius, level, lloyds)  # :69-104
            for parent in self.levels[level - 1]:  # :105-119
                nearby = [c for rn in parent.r_neighbors for c in rn.children]
                for child in parent.children:
                    child.r_neighbors = [c for c in nearby
                                         if _dist(c.point, child.point) <= factor[level] * radius]
            if voronoi:  # :120-158
                for parent in self.levels[level - 1]:
                    self._voronoi(parent)
        self.nodes = [nd for lv in self.levels for nd in lv]

    def _spawn_children(self, parent, radius, level, lloyds):
        x = self.x
        while parent.rows.size > 0:  # :70
            seed = x[parent.rows[0]]  # :71
            point = seed
            if lloyds:  # :72-84
                near = parent.rows[_dist(seed, x[parent.rows]) <= radius]
                point = x[near].mean(axis=-2)
                clash = any(np.linalg.norm(point - c.point) < radius
                            for rn in parent.r_neighbors for c in rn.children)
                if clash:
                    point = seed
            taken = []
            for rn in parent.r_neighbors:  # :89-99
                inside = _dist(point, x[rn.rows]) <= radius
                taken.append(rn.rows[inside])
                rn.rows = rn.rows[~inside]
            child = Node(point, radius, parent, np.concatenate(taken))  # :100
            self.levels[level].append(child)
            parent.children.append(child)

    def _voronoi(self, parent):
        rows = parent.voronoi_rows
        if rows is None or rows.size == 0:  # :123
            return
        nearby = [c for rn in parent.r_neighbors for c in rn.children]  # :124-128
        pts = np.stack([c.point for c in nearby])
        nearest = np.argmin(_dist(pts[:, None, :], self.x[rows][None, :, :]), axis=0)  # :132-135
        for k, child in enumerate(nearby):  # :136-158
            if child.voronoi_rows is None:
                child.voronoi_rows = np.empty((0,), dtype=np.int64)
            child.voronoi_rows = np.concatenate((child.voronoi_rows, rows[nearest == k]))
            child.rows = child.voronoi_rows.copy()

    @property
    def centroids(self):  # :162-164
        return np.stack([nd.point for nd in self.levels[-1]])

    @property
    def cluster_rows(self):
        return [nd.rows for nd in self.levels[-1]]

    @property
    def cluster_mean_and_counts(self):  # :171-179 (an empty cluster has mean nan, count 0)
        means = np.array([np.mean(self.y[nd.rows]) if nd.rows.size else np.nan for nd in self.levels[-1]],
                         dtype=self.y.dtype)
        counts = np.array([nd.rows.size for nd in self.levels[-1]], dtype=self.y.dtype)
        return means[:, None], counts[:, None]


def covertree_update_inducing_parameters(data, spatial_resolution):
    """`cggp/optimize.py:19-38`: centroids, cluster means and counts with empty clusters dropped."""
    tree = CoverTree(data, spatial_resolution=spatial_resolution)
    iv = tree.centroids
    means, counts = tree.cluster_mean_and_counts
    keep = counts.reshape(-1) != 0.0
    return iv[keep], means[keep], counts[keep]
