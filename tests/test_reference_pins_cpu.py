"""The oracle against what the REFERENCE ITSELF computes (the only reference-held pins this tree can have).

Two pieces of the reference depend on the C++ standard library only and are therefore run, not restated
(tests/golden/make_reference_fixtures.py, oracle/Makefile target `ref`, build container only):
  * SolveQuartic / SolveQuarticReals, src/spherical_solvers.cpp:14-98, compiled as they stand;
  * include/RansacLib/{ransac,sampling,utils}.h -- LocallyOptimizedMSAC, UniformSampling, NumRequiredIterations,
    RandomShuffleAndResize -- compiled as they stand and instantiated over the oracle's estimators;
and the generated coefficient arithmetic of both minimal solvers (src/spherical_solvers.cpp:127-277, :338-619) is evaluated
from the reference's own lines by a script.  Their outputs are the committed fixtures tests/golden/ref_*.npz; where oracle/_ref
exists (this container) the comparisons are repeated LIVE on fresh random cases.

Tolerances: integer traces, inlier sets and models of the LO-MSAC comparisons are EQUAL (same estimator code under two drivers);
the quartic restatement is held to 1e-12 relative (it is the same formula through the same libm: observed equal);
the constraint matrices to 1e-12 of their row scale (different association of the same polynomial arithmetic)."""
import os

import numpy as np
import pytest

from spherical_sfm_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
THR = (2 / 600) ** 2


@pytest.fixture(scope="module")
def gold_quartic():
    return np.load(os.path.join(GOLD, "ref_quartic.npz"))


@pytest.fixture(scope="module")
def gold_C():
    return np.load(os.path.join(GOLD, "ref_solver_C.npz"))


@pytest.fixture(scope="module")
def gold_ransaclib():
    return np.load(os.path.join(GOLD, "ref_ransaclib.npz"))


def _have_ref(oracle):
    return oracle.reference_lib_path() is not None


def close_or_both_nan(a, b, rtol):
    a = np.asarray(a); b = np.asarray(b)
    nan = np.isnan(a) | np.isnan(b)
    if (np.isnan(a) != np.isnan(b)).any():
        return False
    scale = max(1.0, np.abs(b[~nan]).max()) if (~nan).any() else 1.0
    return (np.abs(a[~nan] - b[~nan]) <= rtol * scale).all()


# ---- SolveQuartic -------------------------------------------------------------------------------------------------------------------
def test_quartic_restatement_matches_the_compiled_reference(oracle, gold_quartic):
    """oracle solve_quartic == the reference's SolveQuartic (src/spherical_solvers.cpp:15-69) root for root, in its order, incl. the
    complex ones whose real parts the polynomial solver goes on to use (:629-640), near-double roots, the |U.real| < 1e-8 branch (:44-46)
    and biquadratic cases (beta = 0); a NaN / inf on either side would be compared as such."""
    coef, roots = gold_quartic["coef"], gold_quartic["roots"]
    assert len(coef) >= 200
    worst = 0.0; exact = 0
    for c, r in zip(coef, roots):
        mine = oracle.solve_quartic(*c)
        assert close_or_both_nan(mine.real, r.real, 1e-12) and close_or_both_nan(mine.imag, r.imag, 1e-12), (c, mine, r)
        exact += np.array_equal(mine, r, equal_nan=True)
        f = np.isfinite(r)
        if f.any(): worst = max(worst, np.abs(mine[f] - r[f]).max() / max(1.0, np.abs(r[f]).max()))
    print("quartic: %d / %d bit-identical, worst relative difference %.1e" % (exact, len(coef), worst))
    # the branches the fixture is meant to hold (tests/golden/make_reference_fixtures.py: quartic_cases)
    assert (np.abs(roots.imag).max(1) > 1e-3).sum() > 50 and (np.abs(roots.imag).max(1) < 1e-12).sum() > 10


def test_quartic_reals_wrappers(gold_quartic):
    """SolveQuarticReals (:73-98): without tolerance = the real parts of all four roots (what the solver uses), with tolerance = a filter"""
    g = gold_quartic
    n = len(g["reals"])
    assert np.array_equal(g["reals"], g["roots"].real[:n], equal_nan=True)
    for i in range(n):
        keep = g["roots"][i][np.abs(g["roots"][i].imag) < 1e-9].real
        assert g["n_tol"][i] == len(keep) and np.array_equal(g["reals_tol"][i, :len(keep)], keep)


# ---- the generated coefficient matrices --------------------------------------------------------------------------------------------
def test_constraint_matrices_match_the_reference_generated_code(oracle, gold_C):
    """The oracle builds C by polynomial arithmetic from T = 2 E E^T E - tr(E E^T) E (rows -T01, T20, T00, T21, T12, T22); the reference holds
    Matlab-generated expressions.  Same B in, same 6x10 matrix out -- for BOTH layouts (action matrix :127-277, polynomial :338-619)."""
    B = gold_C["B"]
    assert len(B) >= 64
    for variant, key in ((0, "C_action"), (1, "C_poly")):
        worst = 0.0
        for k in range(len(B)):
            mine = oracle.solver_from_basis(B[k], variant)["C"]
            ref = gold_C[key][k]
            err = np.abs(mine - ref).max(1) / np.abs(ref).max(1)
            worst = max(worst, err.max())
        print("variant %d: worst row-relative difference %.1e" % (variant, worst))
        assert worst <= 1e-12


def _match_up_to_sign(Ea, Eb):
    return min(np.abs(Ea - Eb).max(), np.abs(Ea + Eb).max())


def test_polynomial_solver_back_end_matches_reference_chain(oracle, gold_C):
    """C -> G -> quartic a..e (:623-627) -> SolveQuartic -> x (:633-640) -> E (:645-654): the reference chain (its C, numpy's LU, its compiled
    SolveQuartic) against the oracle's from the same B: the quartic's coefficients, all four roots and all four candidates, in order."""
    B = gold_C["B"]
    for k in range(len(B)):
        r = oracle.solver_from_basis(B[k], 1)
        ab = gold_C["abcde"][k]
        assert np.abs(r["abcde"] - ab).max() <= 1e-9 * np.abs(ab).max()            # through a 6x6 solve: conditioning of C[:, :6]
        roots = gold_C["roots_poly"][k]
        # a quartic's roots move with its coefficients by their own conditioning; candidates from well-separated roots are held tightly
        sep = min(abs(roots[i] - roots[j]) for i in range(4) for j in range(i)) / max(1.0, np.abs(roots).max())
        tol = 1e-7 / max(sep, 1e-3)
        for s in range(4):
            assert _match_up_to_sign(r["Es"][s], gold_C["E_poly"][k][s]) <= tol, (k, s, sep)
            assert abs(r["imag"][s] - roots[s].imag) <= tol * max(1.0, abs(roots[s]))


def test_action_matrix_back_end_matches_reference_chain(oracle, gold_C):
    """C -> G -> 4x4 action matrix (:279-285) -> eigenvectors -> E (:289-305).  Eigen's EigenSolver is replaced by numpy's on the reference
    side, so only REAL eigenpairs are comparable (the real part of a complex eigenvector depends on its arbitrary complex scale): every real
    candidate of the reference chain is among the oracle's four, and both sides see the same number of real solutions."""
    B = gold_C["B"]; matched = 0
    for k in range(len(B)):
        r = oracle.solver_from_basis(B[k], 0)
        real = gold_C["E_action_is_real"][k]
        lam = gold_C["lam_action"][k]
        sep = min(abs(lam[i] - lam[j]) for i in range(4) for j in range(i)) / max(1.0, np.abs(lam).max())
        tol = 1e-7 / max(sep, 1e-3)
        for s in np.nonzero(real)[0]:
            d = min(_match_up_to_sign(r["Es"][q], gold_C["E_action"][k][s]) for q in range(4))
            assert d <= tol, (k, s, d, sep)
            matched += 1
    assert matched >= 2 * len(B) * 0.9                                              # real solutions come in pairs; most samples have 2 or 4


def test_reference_candidates_contain_the_true_essential_matrix(gold_C):
    """oracle-free: on the noise-free samples one candidate of the reference chain is the generating E (metric of evaluation/problem_generator.h:17-24)"""
    hits = 0; clean = 0
    for k in range(0, len(gold_C["u"]), 3):                                         # make_reference_fixtures.py: every third sample is noise-free
        u, v, Rgt, Egt, _ = synth.make_relative_pose_problem(3, seed=500 + k, noise=0.0, rotation_deg=3 + k % 50, inward=bool(k % 5 == 4))
        assert np.array_equal(u, gold_C["u"][k])
        Egt = Egt / np.linalg.norm(Egt)
        clean += 1
        for key in ("E_action", "E_poly"):
            hits += min(_match_up_to_sign(E, Egt) for E in gold_C[key][k]) < 1e-8
    assert hits == 2 * clean


# ---- RansacLib ---------------------------------------------------------------------------------------------------------------------
def _pair_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_fixtures", os.path.join(GOLD, "make_reference_fixtures.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_lomsac_restatement_reproduces_the_reference_ransaclib_traces(oracle, gold_ransaclib):
    """oracle/lomsac.hpp against include/RansacLib as compiled: num_iterations, number_lo_iterations, inlier sets, best score and the model, for
    16 option sets x 3 seeds incl. both sampler branches, n < sample size, no consensus, both minimal solvers, LO on / off, final least squares."""
    g = gold_ransaclib; m = _pair_cases()
    ptr = g["pair_ptr"]
    assert len(g["pair_seed"]) == len(m.PAIR_CASES) * len(m.SEEDS)
    lo_total = 0
    for k in range(len(g["pair_seed"])):
        u = g["pair_u"][ptr[k]:ptr[k + 1]]; v = g["pair_v"][ptr[k]:ptr[k + 1]]
        kw = m.PAIR_CASES[g["pair_case"][k]][4]
        r = oracle.lomsac_pair(u, v, THR, seed=int(g["pair_seed"][k]), **kw)
        assert r["iterations"] == g["pair_iterations"][k] and r["lo_runs"] == g["pair_lo_runs"][k], (k, kw)
        assert r["num_inliers"] == g["pair_num_inliers"][k] and np.array_equal(r["inliers"], g["pair_mask"][ptr[k]:ptr[k + 1]])
        assert r["score"] == g["pair_score"][k] and np.array_equal(r["E"], g["pair_E"][k]) and np.array_equal(r["R"], g["pair_R"][k])
        lo_total += int(r["lo_runs"])
    assert lo_total > 50 and g["pair_iterations"].max() > 250 and g["pair_iterations"].min() == 0


def test_retriangulate_restatement_reproduces_the_reference_ransaclib_traces(oracle, gold_ransaclib):
    """SfM::Retriangulate's per-point LocallyOptimizedMSAC<Point, ..., TriangulationEstimator> (src/sfm.cpp:175-183) under the reference's template:
    per point the same iterations, LO runs, inlier flags, zeroing decision and coordinates (tracks of 2..9 observations, corrupted pixels)."""
    g = gold_ransaclib; m = _pair_cases()
    p = m.retriangulate_problem()
    pts, nin, it, lo, fl = oracle.retriangulate_ex(p, num_threads=8)
    assert np.array_equal(it, g["tri_iterations"]) and np.array_equal(lo, g["tri_lo_runs"]) and np.array_equal(nin, g["tri_num_inliers"])
    assert np.array_equal(fl, g["tri_flags"]) and np.array_equal(pts, g["tri_points"])
    assert (lo > 0).sum() > 400 and (pts == 0).all(1).sum() > 10 and (nin < np.bincount(p.obs_pt, minlength=len(pts))).sum() > 20


# ---- live (build container only: oracle/_ref present) ---------------------------------------------------------------------------------
def test_live_random_pairs_under_both_drivers(oracle):
    if not _have_ref(oracle):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine); the committed fixtures above stand in")
    rng = np.random.default_rng(77)
    for k in range(60):
        n = int(rng.choice([3, 4, 6, 10, 30, 100, 250, 600]))
        kw = dict(num_lo_steps=int(rng.integers(0, 12)), num_lsq_iterations=int(rng.choice([0, 2, 4])), final_least_squares=bool(rng.integers(0, 2)),
                  use_poly=bool(rng.integers(0, 2)), lo_starting_iterations=int(rng.choice([0, 5, 50, 120])), max_iterations=int(rng.choice([100, 500, 10000])),
                  success_probability=float(rng.choice([0.9, 0.9999])), seed=int(rng.integers(0, 2 ** 31)))
        if kw["num_lsq_iterations"] == 0: kw["num_lo_steps"] = 0
        u, v, *_ = synth.make_relative_pose_problem(n, seed=9000 + k, noise=float(rng.choice([0, 1 / 600, 3 / 600])), outlier_frac=float(rng.uniform(0, 0.7)),
                                                    rotation_deg=float(rng.uniform(2, 60)))
        a = oracle.lomsac_pair(u, v, THR, **kw)
        with oracle.reference_ransaclib():
            b = oracle.lomsac_pair(u, v, THR, **kw)
        for key in ("iterations", "lo_runs", "num_inliers", "score"):
            assert a[key] == b[key], (k, key, kw)
        assert np.array_equal(a["E"], b["E"]) and np.array_equal(a["inliers"], b["inliers"])


def test_live_quartic_sweep(oracle):
    if not _have_ref(oracle):
        pytest.skip("oracle/_ref not built")
    m = _pair_cases(); L = m.ref_lib()
    rng = np.random.default_rng(5)
    for _ in range(3000):
        c = rng.normal(size=5) * 10.0 ** rng.integers(-3, 4, size=5)
        assert np.array_equal(oracle.solve_quartic(*c), m.ref_quartic(L, c), equal_nan=True), c


# ---- the product's host-side LO-MSAC driver (csrc/shim/lo_msac.h) against the reference's header -------------------------------------------
def test_host_driver_reproduces_the_reference_ransaclib_on_a_toy_estimator(tmp_path):
    """tests/native/lomsac_driver_trace.cpp is written against RansacLib's interface only; compiled against the shim's lo_msac.h it must print, bit for
    bit (%a), what the same source printed when compiled against the reference's include/RansacLib/ransac.h (tests/golden/ref_ransaclib_line.txt:
    11 option sets x 3 seeds of 2-D line fitting: iterations, LO runs, inlier sets, scores, models).  Where /root/reference exists the reference build
    is repeated and compared as well."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "native", "lomsac_driver_trace.cpp")
    gold = open(os.path.join(GOLD, "ref_ransaclib_line.txt")).read()
    assert gold.count("\n") == 33
    builds = [("shim", ["-I" + os.path.join(root, "spherical_sfm_amd", "csrc", "shim")])]
    if os.path.isdir("/root/reference/include/RansacLib"):
        builds.append(("reference", ["-DSSFM_TRACE_REFERENCE_HEADER", "-I/root/reference/include"]))
    for name, flags in builds:
        exe = str(tmp_path / ("trace_" + name))
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off"] + flags + [src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True, check=True).stdout
        assert out == gold, name
