"""Cover-tree clustering (next row F3) -- host mirror of `cggp/covertree.py:13-179`.

Same constructor and properties as the reference class; the construction itself is libmgp's:
`mgp_covertree_build_device` when a GPU is present (the sequential acceptance of centres on the host, the
four all-pairs-shaped passes -- seed balls, the rows a centre takes, Voronoi reassignment, r-neighbour test
-- as device filters over X, `csrc/covertree_dev.hip`), `mgp_covertree_build` (host C++ over row indices,
`csrc/covertree.cpp`) otherwise; the two give the same tree bit for bit.  As in the reference the
`distance` argument is ignored and the Euclidean norm is used (`covertree.py:36-44`).  `data` may be numpy
arrays or torch tensors (the host keeps a copy of x either way, as the reference calls `.numpy()` in
`optimize.py:25`).
"""

import ctypes
import warnings

import numpy as np
import torch

from . import _hip


class CoverTreeNode:
    """Read-only view of one node: `.point [D]`, `.radius`, `.parent` (node or None), `.children`,
    `.rows` (indices of the data rows held) and `.data = (x[rows], y[rows])`."""

    __slots__ = ("point", "radius", "parent", "children", "rows", "_tree")

    def __init__(self, tree, point, radius, parent, rows):
        self._tree, self.point, self.radius, self.parent, self.rows = tree, point, radius, parent, rows
        self.children = []

    @property
    def data(self):
        return self._tree._x[self.rows], self._tree._y[self.rows]


def _host(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


class CoverTree:
    def __init__(self, distance, data, spatial_resolution=None, num_levels=1, lloyds=True, voronoi=True,
                 plotting=False, *, device=None):
        """`device`: None = the GPU-assisted construction whenever a GPU is there (on the device of `data[0]` if
        that is a CUDA tensor, else the current one), False = the host construction, a torch device = that GPU."""
        warnings.warn("Distance function will be ignored and instead the Euclidean norm will be used.")
        dev = None
        if device is not False and torch.cuda.is_available():
            if isinstance(device, (torch.device, str)):
                dev = torch.device(device)
            elif isinstance(data[0], torch.Tensor) and data[0].is_cuda:
                dev = data[0].device
            else:
                dev = torch.device("cuda", torch.cuda.current_device())
        elif device not in (None, False):
            raise RuntimeError("CoverTree(device=...) needs a GPU")
        x_in = data[0]
        x, y = (_host(a) for a in data)
        if x.ndim != 2 or y.ndim != 2 or x.shape[0] != y.shape[0]:
            raise ValueError("data must be (x [N,D], y [N,Dy])")
        self._x, self._y = x, y
        x64 = np.ascontiguousarray(x, dtype=np.float64)
        lib = _hip.load_library()
        handle = ctypes.c_void_p()
        res = 0.0 if spatial_resolution is None else float(spatial_resolution)
        if spatial_resolution is not None and not res > 0.0:
            raise ValueError("spatial_resolution must be positive")
        if dev is not None:
            hd = _hip.get_handle(dev)
            if isinstance(x_in, torch.Tensor) and x_in.is_cuda and x_in.device == dev:
                x_dev = x_in.detach().to(torch.float64).contiguous()
            else:
                x_dev = torch.from_numpy(x64).to(dev)
            rc = lib.mgp_covertree_build_device(hd.h, x64.ctypes.data, _hip.ptr(x_dev), x.shape[0], x.shape[1], res,
                                                int(num_levels or 1), int(bool(lloyds)), int(bool(voronoi)),
                                                ctypes.byref(handle))
        else:
            rc = lib.mgp_covertree_build(x64.ctypes.data, x.shape[0], x.shape[1], res, int(num_levels or 1),
                                         int(bool(lloyds)), int(bool(voronoi)), ctypes.byref(handle))
        self.built_on = "device" if dev is not None else "host"
        if rc != 0:
            raise RuntimeError(f"mgp_covertree_build failed ({rc}): {lib.mgp_host_last_error().decode()}")
        try:
            self.levels = []
            for level in range(lib.mgp_covertree_num_levels(handle)):
                n = lib.mgp_covertree_level_size(handle, level)
                pts = np.empty((n, x.shape[1]), dtype=np.float64)
                parent = np.empty((n,), dtype=np.int64)
                counts = np.empty((n,), dtype=np.int64)
                offsets = np.empty((n + 1,), dtype=np.int64)
                lib.mgp_covertree_level_nodes(handle, level, pts.ctypes.data, parent.ctypes.data, counts.ctypes.data)
                rows = np.empty((int(counts.sum()),), dtype=np.int64)
                lib.mgp_covertree_level_rows(handle, level, offsets.ctypes.data, rows.ctypes.data)
                radius = lib.mgp_covertree_level_radius(handle, level)
                pts = pts.astype(x.dtype, copy=False)
                nodes = []
                for k in range(n):
                    up = self.levels[level - 1][parent[k]] if level > 0 else None
                    nd = CoverTreeNode(self, pts[k], radius, up, rows[offsets[k]:offsets[k + 1]])
                    if up is not None:
                        up.children.append(nd)
                    nodes.append(nd)
                self.levels.append(nodes)
                if level == lib.mgp_covertree_num_levels(handle) - 1:
                    self._leaf = (pts, offsets, rows)
        finally:
            lib.mgp_covertree_destroy(handle)
        self.nodes = [nd for lv in self.levels for nd in lv]

    @property
    def centroids(self):
        """`covertree.py:161-163`: centres of the finest level [K, D]."""
        return self._leaf[0]

    @property
    def cluster_ys(self):
        """`:165-168`."""
        return [self._y[nd.rows] for nd in self.levels[-1]]

    @property
    def cluster_mean_and_counts(self):
        """`:170-179`: (means [K,1], counts [K,1]) in y's dtype; an empty cluster has mean NaN and
        count 0 (numpy's mean of nothing), which `covertree_update_inducing_parameters` filters."""
        _, offsets, rows = self._leaf
        y = self._y.astype(np.float64, copy=False)
        yv = y[rows].reshape(rows.shape[0], -1)
        counts = np.diff(offsets)
        per_row = yv.shape[1] if yv.ndim > 1 else 1
        sums = np.zeros((counts.shape[0],), dtype=np.float64)
        nz = counts > 0
        if rows.shape[0]:
            sums[nz] = np.add.reduceat(yv.sum(axis=1), offsets[:-1][nz])
        with np.errstate(invalid="ignore", divide="ignore"):
            means = sums / (counts * per_row)
        return means.astype(self._y.dtype)[:, None], counts.astype(self._y.dtype)[:, None]
