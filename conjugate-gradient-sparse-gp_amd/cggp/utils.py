"""`cggp/utils.py:11-17`."""

import torch


def add_diagonal(matrix, diagonal):
    """Returns `matrix + diag(diagonal)` for a [n,n] matrix and a length-n vector (not in place)."""
    if matrix.dim() != 2 or matrix.shape[0] != matrix.shape[1]:
        raise ValueError("matrix must be [n, n]")
    diagonal = diagonal.reshape(-1)
    if diagonal.shape[0] != matrix.shape[0]:
        raise ValueError("diagonal must have n entries")
    out = matrix.clone()
    out.diagonal().add_(diagonal.to(out.dtype))
    return out


# ------------------------------------------------------------------ row F4: parameters and reports
def parameter_dict(model):
    """GPflow-path keyed parameter dict, the format the reference stores as `params.npy`
    (`cggp/utils.py:29-38`, `cggp/paper_cli_geospatial.py:299-301`): `.kernel.variance`,
    `.kernel.lengthscales`, `.likelihood.variance`, `.inducing_variable.Z`, `.pseudo_u`,
    `.cluster_counts` (numpy values)."""
    import numpy as np

    def arr(x):
        return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x, dtype=np.float64)

    out = {
        ".kernel.variance": arr(model.kernel.variance),
        ".kernel.lengthscales": arr(model.kernel.lengthscales),
        ".likelihood.variance": arr(model.likelihood.variance),
        ".inducing_variable.Z": arr(model.inducing_variable.Z),
    }
    if hasattr(model, "pseudo_u"):
        out[".pseudo_u"] = arr(model.pseudo_u)
        out[".cluster_counts"] = arr(model.cluster_counts)
    return out


def multiple_assign(model, params):
    """`gpflow.utilities.multiple_assign` for the keys above (`cggp/paper_cli_uci.py:123-124`)."""
    Z = model.inducing_variable.Z
    for key, val in params.items():
        if key == ".kernel.variance":
            model.kernel.variance = float(val)
        elif key == ".kernel.lengthscales":
            import numpy as np
            model.kernel.lengthscales = [float(v) for v in np.atleast_1d(val)]
        elif key == ".likelihood.variance":
            model.likelihood.variance = float(val)
        elif key == ".inducing_variable.Z":
            model.inducing_variable.Z = torch.as_tensor(val, dtype=Z.dtype, device=Z.device)
        elif key == ".pseudo_u":
            model.pseudo_u = torch.as_tensor(val, dtype=Z.dtype, device=Z.device)
        elif key == ".cluster_counts":
            model.cluster_counts = torch.as_tensor(val, dtype=Z.dtype, device=Z.device)
        else:
            raise KeyError(f"unknown parameter path {key}")


def store_params(path, params):
    """Parameter dict -> `.npz` (plain arrays, no pickle).  The reference writes a pickled dict
    with `np.save(allow_pickle=True)` (`cggp/utils.py:29-32`); reading such files means unpickling,
    so this build reads and writes the pickle-free form only."""
    import os
    import numpy as np
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    np.savez(path, **{k: np.asarray(v) for k, v in params.items()})


def load_params(path):
    import numpy as np
    with np.load(path, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def store_reference_params(path, params):
    """Write the reference's own `params.npy` layout (`cggp/utils.py:29-32`: one pickled dict of
    name -> ndarray inside a 0-d object array) so its `load_from_npy` reads our parameters."""
    import os
    import numpy as np
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    np.save(path, {str(k): np.asarray(v) for k, v in params.items()}, allow_pickle=True)


def load_reference_params(path):
    """Read a `params.npy` written by the reference (`np.save(path, dict, allow_pickle=True)`)
    WITHOUT general unpickling: the `.npy` header is parsed by numpy, and the payload goes through
    an unpickler whose `find_class` admits only what a dict of plain ndarrays needs (numpy array /
    dtype / scalar reconstruction, `_codecs.encode`) -- any other global in the stream raises, so
    nothing from the file is ever called."""
    import pickle
    import numpy as np

    allowed = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode"),
    }

    class _ArraysOnly(pickle.Unpickler):
        def find_class(self, module, name):
            if (module, name) not in allowed:
                raise pickle.UnpicklingError(f"refusing {module}.{name}: not a plain-array pickle")
            if module == "_codecs":
                import _codecs
                return _codecs.encode
            if "multiarray" in module:
                try:
                    import numpy._core.multiarray as ma
                except ImportError:  # numpy < 2
                    import numpy.core.multiarray as ma
                return getattr(ma, name)
            return getattr(np, name)

    with open(path, "rb") as fh:
        version = np.lib.format.read_magic(fh)
        if version == (1, 0):
            shape, _, dtype = np.lib.format.read_array_header_1_0(fh)
        else:
            shape, _, dtype = np.lib.format.read_array_header_2_0(fh)
        if dtype.hasobject:
            if shape != ():
                raise ValueError("expected a 0-d object array holding one dict")
            obj = _ArraysOnly(fh).load()
        else:
            raise ValueError("not a pickled parameter dict; use numpy.load for plain arrays")
    if isinstance(obj, np.ndarray) and obj.dtype.hasobject and obj.shape == ():
        obj = obj.item()  # np.save pickles the 0-d object array that wraps the dict
    if not isinstance(obj, dict):
        raise ValueError("params file does not hold a dict")
    out = {}
    for k, v in obj.items():
        v = np.asarray(v)
        if v.dtype.hasobject:
            raise ValueError(f"parameter {k!r} is not a plain array")
        out[str(k)] = v
    return out


def covariance_properties(model, jitter=0.0, with_lambda=False):
    """Condition-number report (`cggp/paper_cli_uci.py:174-185`; with `with_lambda` the
    `Kuu + Lambda` form of `cggp/paper_condition_wasserstein.py:115-124`).  The symmetric
    eigen-decomposition is a plain library call on the [M,M] matrix libmgp builds."""
    from .kernels import Kuu
    diag = model.diag_variance[:, 0] if with_lambda else None
    K = Kuu(model.inducing_variable, model.kernel, jitter=jitter, diag_add=diag)
    ev = torch.linalg.eigvalsh(K)
    eig_min, eig_max = float(ev.min()), float(ev.max())
    return dict(condition_number=eig_max / eig_min, eig_min=eig_min, eig_max=eig_max)
