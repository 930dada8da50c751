"""The HIP path (through the C ABI) against fixtures that come from the REFERENCE ITSELF (tests/golden/ref_*.npz, produced in the build
container by tests/golden/make_reference_fixtures.py; nothing here reads /root/reference):

  * ref_solver_C.npz  -- candidate essential matrices from the reference's generated coefficient code (src/spherical_solvers.cpp:127-277,
                         :338-619, evaluated as written) + its compiled SolveQuartic (:14-98): the device's two minimal solvers (row a11) must
                         return the same real candidates for the same three rays (E is basis independent; <= 1e-7 / root separation).
  * ref_ransaclib.npz -- RansacStatistics / inlier sets / models produced by the reference's own include/RansacLib template: the device's
                         reference-trace LO-MSAC (row a13, k_lomsac_trace) and Retriangulate (row N1, k_retriangulate_trace) must show the same
                         num_iterations, number_lo_iterations and inlier flags.  EQUALITY is required wherever the floating-point work
                         between two decisions is well conditioned (estimate_pairwise's own options, every Retriangulate point); with in-loop
                         local optimisation the ill-conditioned non-minimal solves may part the runs (tests/test_ransac_trace_gpu.py shows where)
                         and a fraction is required instead.
Tolerances as stated per test; north_star asks for <= 1e-5 relative pose error."""
import importlib.util
import os

import numpy as np
import pytest

from spherical_sfm_amd import ba, ransac
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
THR = (2 / 600) ** 2


def _fixture_module():
    spec = importlib.util.spec_from_file_location("make_reference_fixtures", os.path.join(GOLD, "make_reference_fixtures.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _sign_dist(Ea, Eb):
    return min(np.abs(Ea - Eb).max(), np.abs(Ea + Eb).max())


@pytest.mark.parametrize("poly", [False, True])
def test_minimal_solvers_return_the_reference_chain_candidates(gpu_ctx, poly):
    g = np.load(os.path.join(GOLD, "ref_solver_C.npz"))
    S = len(g["u"])
    u = g["u"].reshape(-1, 3); v = g["v"].reshape(-1, 3)
    samples = np.arange(3 * S, dtype=np.int32).reshape(S, 3)
    out = ransac.solver_probe(gpu_ctx, u, v, samples, poly=poly)
    checked = 0; worst = 0.0
    for k in range(S):
        if poly:
            roots = g["roots_poly"][k]; real = np.abs(roots.imag) <= 1e-9 * np.maximum(1.0, np.abs(roots)); Eref = g["E_poly"][k]; z = roots
        else:
            real = g["E_action_is_real"][k]; Eref = g["E_action"][k]; z = g["lam_action"][k]
        sep = min(abs(z[i] - z[j]) for i in range(4) for j in range(i)) / max(1.0, np.abs(z).max())
        tol = 1e-7 / max(sep, 1e-3)
        assert len(out[k]) >= real.sum(), (k, len(out[k]), real.sum())
        for s in np.nonzero(real)[0]:
            d = min(_sign_dist(E, Eref[s]) for E in out[k])
            assert d <= tol, (k, s, d, sep)
            worst = max(worst, d); checked += 1
    print("real candidates checked: %d, worst |dE| %.2e" % (checked, worst))
    assert checked >= 1.8 * S * 0.9


_KW = dict(use_poly="use_poly_solver", max_iterations="max_num_iterations", min_iterations="min_num_iterations")


def test_lomsac_trace_equals_the_reference_ransaclib_runs(gpu_ctx):
    g = np.load(os.path.join(GOLD, "ref_ransaclib.npz")); m = _fixture_module()
    ptr = g["pair_ptr"]
    plain, lo, parting = [], [], []
    for k in range(len(g["pair_seed"])):
        kw = dict(m.PAIR_CASES[g["pair_case"][k]][4])
        dev = {_KW.get(a, a): (int(b) if isinstance(b, (bool, np.bool_)) else b) for a, b in kw.items()}
        dev.setdefault("final_least_squares", 1); dev.setdefault("num_lo_steps", 0); dev.setdefault("num_lsq_iterations", 0)
        u = g["pair_u"][ptr[k]:ptr[k + 1]]; v = g["pair_v"][ptr[k]:ptr[k + 1]]
        out = ransac.estimate_pairs(gpu_ctx, [(u, v)], THR, seed=int(g["pair_seed"][k]), min_num_inliers=0, **dev)
        same = (out["iterations"][0] == g["pair_iterations"][k] and out["lo_runs"][0] == g["pair_lo_runs"][k]
                and out["num_inliers"][0] == g["pair_num_inliers"][k] and np.array_equal(out["inliers"][0], g["pair_mask"][ptr[k]:ptr[k + 1]]))
        # (the model itself is compared where the data determine it: with 3..5 rays the six-parameter least-squares fit that ends the run is under-determined and
        #  its answer is set by the Levenberg-Marquardt damping alone -- there the trace, the inlier flags and the score are the statement)
        close = (not same) or g["pair_num_inliers"][k] == 0 or len(u) <= 5 or _sign_dist(out["E"][0], g["pair_E"][k]) <= 1e-8
        if same and len(u) <= 5 and g["pair_num_inliers"][k] > 0:
            # (round 6) ... and what IS determined there is compared: the reference's model fits its 3..5 inlier rays exactly (summed Sampson error
            # <= 2e-30 in the fixture), so must the device's -- whichever member of the solution family the damping picked (scripts/r06/small_pairs.py prints both)
            mk = g["pair_mask"][ptr[k]:ptr[k + 1]].astype(bool)
            fit = lambda E: sum(O.sampson(E, u[i], v[i]) for i in range(len(u)) if mk[i])
            assert fit(g["pair_E"][k]) <= 1e-28 and fit(out["E"][0]) <= 1e-28, (k, fit(out["E"][0]), fit(g["pair_E"][k]))
        (lo if dev["num_lo_steps"] > 0 else plain).append(same and close)
        if dev["num_lo_steps"] > 0 and not (same and close):
            parting.append(k)
        if dev["num_lo_steps"] == 0:
            assert same and close, (k, kw, out["iterations"][0], g["pair_iterations"][k], out["num_inliers"][0], g["pair_num_inliers"][k])
    print("estimate_pairwise-style options: %d / %d identical; with in-loop local optimisation: %d / %d" % (sum(plain), len(plain), sum(lo), len(lo)))
    # round 6 (VERDICT r5 #7a): the runs with in-loop local optimisation that leave the reference's are LISTED, not bounded by a rate: of the 36 such runs one does --
    # run 4 (500 correspondences, 100 iterations: 337 against 338 inliers after an ill-conditioned non-minimal solve picked the other of two near-equal candidates).
    # A second parting run is a regression; run 4 starting to agree is welcome (scripts/r06/list_parting.py prints the list).
    assert all(plain) and set(parting) <= {4}, parting


def test_retriangulate_trace_equals_the_reference_ransaclib_runs(gpu_ctx):
    g = np.load(os.path.join(GOLD, "ref_ransaclib.npz")); m = _fixture_module()
    p = m.retriangulate_problem()
    X, nin, it, lo, fl = ba.retriangulate_ex(gpu_ctx, p)
    assert np.array_equal(it, g["tri_iterations"]) and np.array_equal(lo, g["tri_lo_runs"]) and np.array_equal(nin, g["tri_num_inliers"])
    assert np.array_equal(fl, g["tri_flags"])
    Xr = g["tri_points"]; zero = ~Xr.any(1)
    assert np.array_equal(~X.any(1), zero)
    rel = np.linalg.norm(X - Xr, axis=1)[~zero] / np.linalg.norm(Xr[~zero], axis=1)
    assert rel.max() <= 1e-9
