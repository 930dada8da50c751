"""The oracle's restatement of Ceres' projected Armijo line search (oracle/line_search.hpp) against numpy: the interpolating polynomial
and its minimiser, the real parts of polynomial roots, and the search itself on functions with a known answer.  (The product's own
copy, csrc/line_search.h, is host code of libssfm_hip.so and is compared with this one through the pose-graph solves on the GPU,
tests/test_rotavg_gpu.py.)"""
import ctypes as C
import numpy as np
import pytest


@pytest.fixture(scope="module")
def L(oracle):
    lib = oracle.lib()
    dp = C.POINTER(C.c_double)
    lib.oracle_ls_polynomial.argtypes = [C.c_int32, dp, dp, dp, C.POINTER(C.c_uint8), dp]; lib.oracle_ls_polynomial.restype = C.c_int32
    lib.oracle_ls_roots.argtypes = [C.c_int32, dp, dp]; lib.oracle_ls_roots.restype = C.c_int32
    lib.oracle_ls_step.argtypes = [dp, C.c_double, C.c_double]; lib.oracle_ls_step.restype = C.c_double
    lib.oracle_ls_armijo_poly.argtypes = [C.c_int32, dp, C.c_double, dp, C.POINTER(C.c_int32)]; lib.oracle_ls_armijo_poly.restype = C.c_int32
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_interpolating_polynomial_reproduces_its_data(L):
    rng = np.random.default_rng(0)
    for trial in range(50):
        ns = int(rng.integers(2, 4))
        x = np.sort(rng.uniform(0, 1.5, ns)); x[0] = 0.0
        true = rng.normal(size=2 * ns)                                  # a polynomial of exactly the interpolant's degree
        f = np.polyval(true, x); df = np.polyval(np.polyder(true), x)
        flags = np.ones(2 * ns, np.uint8); out = np.zeros(6)
        m = L.oracle_ls_polynomial(ns, _dp(x), _dp(f), _dp(df), flags.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(out))
        assert m == 2 * ns and np.allclose(out[:m], true, rtol=1e-7, atol=1e-7 * np.abs(true).max())


def test_root_real_parts_match_numpy(L):
    rng = np.random.default_rng(1)
    for deg in (1, 2, 3, 4):
        for trial in range(40):
            c = rng.normal(size=deg + 1)
            if trial % 5 == 0 and deg >= 2:
                c = np.poly(rng.normal(size=deg))                       # all roots real
            out = np.zeros(5); k = L.oracle_ls_roots(deg, _dp(np.ascontiguousarray(c)), _dp(out))
            assert k == deg
            assert np.allclose(np.sort(out[:k]), np.sort(np.roots(c).real), rtol=1e-8, atol=1e-8)
    out = np.zeros(5)
    assert L.oracle_ls_roots(3, _dp(np.array([0.0, 0.0, 2.0, -3.0])), _dp(out)) == 1 and abs(out[0] - 1.5) < 1e-15     # leading zeros removed


def test_step_size_minimises_the_cubic_through_two_samples(L):
    rng = np.random.default_rng(2)
    for trial in range(100):
        f0 = rng.uniform(1, 2); g0 = -rng.uniform(0.1, 2); f1 = f0 + rng.uniform(-0.1, 1.0); g1 = rng.normal()
        s = np.array([f0, g0, 1.0, f1, g1])
        a = L.oracle_ls_step(_dp(s), 1e-3, 0.6)
        # the Hermite cubic on [0, 1] and a dense search over [1e-3, 0.6]
        A = np.array([[0, 0, 0, 1], [0, 0, 1, 0], [1, 1, 1, 1], [3, 2, 1, 0]], float)
        c = np.linalg.solve(A, [f0, g0, f1, g1])
        grid = np.linspace(1e-3, 0.6, 200001)
        assert np.polyval(c, a) <= np.polyval(c, grid).min() + 1e-12 and 1e-3 <= a <= 0.6


def test_armijo_on_polynomials(L):
    """f(a) = p(a): the search returns a step with sufficient decrease, 1 when the full step already has it, and gives up on a
    direction that is not a descent direction within its 20 iterations."""
    a = C.c_double(0); n = C.c_int32(0)
    p = np.array([1.0, -1.0, 0.0])                                       # a^2 - a: minimum at 0.5, f(1) = f(0)
    assert L.oracle_ls_armijo_poly(2, _dp(p), 1.0, C.byref(a), C.byref(n)) == 1
    assert 0.0 < a.value < 1.0 and np.polyval(p, a.value) <= 0 + 1e-4 * (-1.0) * a.value and abs(a.value - 0.5) < 1e-12     # the cubic fit is exact
    p = np.array([0.1, -1.0, 3.0])
    assert L.oracle_ls_armijo_poly(2, _dp(p), 1.0, C.byref(a), C.byref(n)) == 1 and a.value == 1.0 and n.value == 1
    p = np.array([5.0, -4.0, 1.0, -0.01, 2.0])                           # steep walls: needs several contractions
    assert L.oracle_ls_armijo_poly(4, _dp(p), 1.0, C.byref(a), C.byref(n)) == 1 and n.value >= 2
    assert np.polyval(p, a.value) <= 2.0 + 1e-4 * (-0.01) * a.value
    p = np.array([1.0, 0.5, 0.0])                                        # uphill: slope +0.5 at 0
    assert L.oracle_ls_armijo_poly(2, _dp(p), 1.0, C.byref(a), C.byref(n)) == 0
