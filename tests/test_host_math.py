"""csrc/mgp_math.h compiled for the host: the exp2 forms and the kernel profiles the HIP kernels
inline are checked against libm / the oracle on the CPU (the header is shared, device and host)."""

import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle.kernels import Kernel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd", "csrc")
SO = os.path.join(ROOT, "conjugate-gradient-sparse-gp_amd", "cggp", "libmgp_hostmath.so")


@pytest.fixture(scope="module")
def hm():
    # MGP_HOSTMATH_LIBRARY: the sanitizer run (tests/test_sanitizers.py) points this at the ASan/UBSan build
    so = os.environ.get("MGP_HOSTMATH_LIBRARY")
    if not so:
        subprocess.run(["make", "-C", CSRC, "hostmath"], check=True, capture_output=True)
        so = SO
    lib = ctypes.CDLL(so)
    for n in ("mgp_host_exp2", "mgp_host_exp2_tab"):
        getattr(lib, n).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    lib.mgp_host_exp2_shifted.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_long]
    lib.mgp_host_profile.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long]
    lib.mgp_host_profile_scale.restype = ctypes.c_double
    return lib


def _args():
    rng = np.random.default_rng(0)
    return np.concatenate([-rng.random(200000) * 60, np.linspace(-1080, 0.5, 100001),
                           [0.0, -0.5, -1.5, -1074.0, -1100.0, -5000.0, -1e9, -1e300, 1e-17, 3.0]])


@pytest.mark.parametrize("fn", ["mgp_host_exp2", "mgp_host_exp2_tab"])
def test_exp2_within_one_ulp(hm, fn):
    t = _args()
    out = np.empty_like(t)
    getattr(hm, fn)(t.ctypes.data, out.ctypes.data, t.size)
    ref = np.exp2(t)
    m = ref > 1e-300
    assert np.max(np.abs(out[m] / ref[m] - 1)) <= 2.3e-16
    assert np.all(out[~m] <= 1e-299) and np.all(out >= 0) and np.all(np.isfinite(out))
    one = np.zeros(1)
    getattr(hm, fn)(one.ctypes.data, one.ctypes.data, 1)
    assert one[0] == 1.0  # 2^0 exactly: k(x,x) == variance when r2 == 0


@pytest.mark.parametrize("kind,name", list(enumerate(["se", "matern12", "matern32", "matern52"])))
def test_profiles_match_gpflow_formulas(hm, kind, name):
    r2 = np.concatenate([np.random.default_rng(1).random(100000) * 50, [0.0, 1e-40, 1.0]])
    c = hm.mgp_host_profile_scale(kind)
    s = r2 * c * c
    out = np.empty_like(s)
    hm.mgp_host_profile(kind, s.ctypes.data, out.ctypes.data, s.size)
    ref = Kernel(name).K_r2(r2)
    assert np.max(np.abs(out - ref) / ref) < 2e-14
    assert out[-3] == 1.0 and abs(out[-2] - 1.0) < 1e-15


def test_shifted_table_form_of_the_se_sweep(hm):
    """2^(s - a2) with a2 folded into the magic constant (the SE fast path of csrc/sweep.hip)."""
    rng = np.random.default_rng(0)
    n = 300000
    a2 = rng.random(n) * rng.choice([1, 10, 1000, 2e5], n)
    s = -rng.random(n) * rng.choice([1, 60, 1000], n) + a2
    out = np.empty(n)
    hm.mgp_host_exp2_shifted(s.ctypes.data, a2.ctypes.data, out.ctypes.data, n)
    ref = np.exp2(s.astype(np.longdouble) - a2.astype(np.longdouble)).astype(np.float64)
    m = ref > 1e-300
    assert np.max(np.abs(out[m] / ref[m] - 1)) <= 5e-16
