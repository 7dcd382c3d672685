"""The tiled multi-workgroup Cholesky (fp64 MFMA tile updates) behind windows with D > 174, stand-alone:
okvis_ba_dense_solve against numpy on SPD systems of the sizes the path sees (BASELINE configs[2]: D = 750)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _solve(S, r):
    from okvis_amd import _lib
    L = _lib.lib()
    n = S.shape[0]
    S = np.ascontiguousarray(S, np.float64)
    r = np.ascontiguousarray(r, np.float64)
    x = np.zeros(n)
    info = C.c_int32(-1)
    dp = C.POINTER(C.c_double)
    st = L.okvis_ba_dense_solve(0, n, S.ctypes.data_as(dp), r.ctypes.data_as(dp), x.ctypes.data_as(dp), C.byref(info))
    assert st == 0, st
    return x, info.value


@pytest.mark.parametrize("n", [1, 47, 48, 49, 150, 750, 900])
def test_spd_systems(n):
    rng = np.random.default_rng(n)
    A = rng.normal(size=(n, n))
    S = A @ A.T + n * np.eye(n)
    S[np.arange(n), np.arange(n)] *= 1 + rng.uniform(0, 3, n)      # asymmetric scaling of rows/cols is kept symmetric
    S = 0.5 * (S + S.T)
    r = rng.normal(size=n)
    x, info = _solve(S, r)
    assert info == 0
    ref = np.linalg.solve(S, r)
    assert np.abs(x - ref).max() <= 1e-10 * np.abs(ref).max()


def test_graded_system_like_the_reduced_camera_matrix():
    rng = np.random.default_rng(5)
    n = 750
    A = rng.normal(size=(n, n)) * 0.01
    d = 10.0 ** rng.uniform(0, 8, n)           # 1e8 dynamic range on the diagonal (priors vs weak directions)
    S = A @ A.T + np.diag(d)
    r = rng.normal(size=n) * np.sqrt(d)
    x, info = _solve(S, r)
    assert info == 0
    ref = np.linalg.solve(S, r)
    assert np.abs(x - ref).max() <= 1e-9 * np.abs(ref).max()


def test_not_positive_definite_is_reported():
    n = 100
    S = np.eye(n)
    S[60, 60] = -1.0
    x, info = _solve(S, np.ones(n))
    assert info == 1 and np.all(np.isfinite(x))
