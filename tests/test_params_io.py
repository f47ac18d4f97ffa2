"""Row F4: parameter files in the reference's own layout (`cggp/utils.py:29-38`).

The reference writes `np.save(path, dict_of_arrays, allow_pickle=True)`.  `load_reference_params`
reads that layout through an unpickler that admits only plain-array reconstruction; the files
used here are written by this test (nothing shipped with the reference is loaded).
"""
import pickle

import numpy as np
import pytest


def test_reference_params_round_trip(tmp_path):
    from cggp.utils import load_reference_params, store_reference_params
    params = {".kernel.variance": np.float64(1.3), ".kernel.lengthscales": np.array([0.5, 2.0, 1.0]),
              ".likelihood.variance": np.array(0.1), ".inducing_variable.Z": np.arange(12.0).reshape(4, 3),
              ".cluster_counts": np.array([[3.0], [1.0], [7.0], [2.0]], dtype=np.float32)}
    path = tmp_path / "params.npy"
    np.save(path, params, allow_pickle=True)  # exactly what the reference's store_as_npy does
    got = load_reference_params(path)
    assert set(got) == set(params)
    for k in params:
        assert np.array_equal(got[k], np.asarray(params[k])) and got[k].dtype == np.asarray(params[k]).dtype
    # and the writer produces what the reference's reader expects: np.load(...).item() is the dict
    out = tmp_path / "ours.npy"
    store_reference_params(out, got)
    back = np.load(out, allow_pickle=True).item()  # our own file
    assert set(back) == set(params) and all(np.array_equal(back[k], got[k]) for k in got)


class _Evil:
    def __reduce__(self):
        import os
        return (os.system, ("echo pwned > /dev/null",))


def test_reference_params_refuses_anything_but_arrays(tmp_path):
    from cggp.utils import load_reference_params
    path = tmp_path / "evil.npy"
    np.save(path, {"a": np.zeros(2), "b": _Evil()}, allow_pickle=True)
    with pytest.raises(pickle.UnpicklingError):
        load_reference_params(path)
    plain = tmp_path / "plain.npy"
    np.save(plain, np.zeros(3))
    with pytest.raises(ValueError):
        load_reference_params(plain)
    arr = tmp_path / "objarr.npy"
    np.save(arr, np.array([{"a": 1}, {"b": 2}], dtype=object), allow_pickle=True)
    with pytest.raises(ValueError):
        load_reference_params(arr)
