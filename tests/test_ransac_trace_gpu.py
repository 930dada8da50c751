"""GPU parity of the reference-trace LO-MSAC and of its pieces against the oracle, through the C ABI.

In this mode the device restates RansacLib's control flow draw for draw (std::mt19937 + libstdc++'s uniform_int_distribution, both
streams), so -- unlike the fixed-budget mode of tests/test_ransac_gpu.py, which is compared statistically -- every comparison here is
deterministic: same iteration counts, same number of LocalOptimization runs, same inlier flags, E / R to rounding.
Tolerances: E (unit Frobenius norm, up to sign) and R within 1e-9 of the oracle's for a pair, 1e-9 for the single-function probes
(north_star asks for <= 1e-5 relative pose error)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from spherical_sfm_amd import synth

pytestmark = pytest.mark.gpu
THR = (2 / 600) ** 2


def frob_err(E, Es):
    a = E / np.linalg.norm(E); b = Es / np.linalg.norm(Es)
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def rot_err(R, Rs):
    return np.linalg.norm(Rotation.from_matrix(Rs @ R.T).as_rotvec())


def _pairs(n_pairs, n_corr, outlier_frac, noise, seed0=100):
    return [synth.make_relative_pose_problem(n_corr, seed=seed0 + k, noise=noise, outlier_frac=outlier_frac, rotation_deg=5 + (k % 30))
            for k in range(n_pairs)]


# ---- the random streams ------------------------------------------------------------------------------------------------------
def test_mt19937_and_uniform_int_match_libstdcxx_and_numpy(gpu_ctx, oracle):
    from spherical_sfm_amd import ransac
    rng = np.random.default_rng(1)
    n = 3000                                                   # crosses several 624-word state transitions
    hi = rng.integers(0, 5000, n).astype(np.int32); lo = (hi * rng.uniform(0, 1, n)).astype(np.int32)
    hi[:5] = [0, 1, 2, 2 ** 31 - 1, 2 ** 30 + 12345]; lo[:5] = 0      # range 1, 2, 3 and two huge ranges (Lemire rejections are likely there)
    for seed in (0, 1, 5489, 2 ** 32 - 1):
        raw_g, d_g = ransac.mt19937_probe(gpu_ctx, seed, lo, hi, nraw=700)
        raw_o, d_o = oracle.mt19937_draws(seed, lo, hi, nraw=700)
        assert (raw_g == raw_o).all() and (d_g == d_o).all()
        # third-party anchor: numpy's legacy generator is init_genrand(seed) + the same tempering
        assert (raw_g == np.random.RandomState(seed).randint(0, 2 ** 32, 700, dtype=np.uint64).astype(np.uint32)).all()


# ---- the estimator's pieces, one function at a time (row a12) -------------------------------------------------------------------
@pytest.mark.parametrize("inward", [False, True])
def test_least_squares_probe_matches_oracle(gpu_ctx, oracle, inward):
    """SphericalEstimator::LeastSquares: same start model + same ray subset -> the same E (src/spherical_estimator.cpp:110-157)."""
    from spherical_sfm_amd import ransac
    u, v, R, E, inl = synth.make_relative_pose_problem(300, seed=11, noise=1 / 600, outlier_frac=0.25, rotation_deg=14, inward=inward)
    rng = np.random.default_rng(3)
    good = np.nonzero(inl)[0]
    lists, starts = [], []
    for k in range(24):
        m = [21, 21, 7, 3, 60, len(good)][k % 6]
        lists.append(rng.choice(good, m, replace=False).astype(np.int32))
        # start models: minimal-solver solutions of a clean sample (what RANSAC hands over), the ground truth, and perturbed truths
        if k % 3 == 0:
            cands = oracle.spherical_solver(u, v, rng.choice(good, 3, replace=False).astype(np.int32))
            starts.append(min(cands, key=lambda e: frob_err(e, E)))
        elif k % 3 == 1:
            starts.append(E / np.linalg.norm(E))
        else:
            Rp = Rotation.from_rotvec(rng.normal(size=3) * 0.02).as_matrix() @ R
            starts.append(oracle.make_spherical_essential_matrix(Rp, inward))
    got = ransac.sampson_refine_probe(gpu_ctx, u, v, lists, starts, inward=inward)
    refs = [oracle.sampson_least_squares_ex(u, v, lst, e0, inward=inward) for lst, e0 in zip(lists, starts)]
    worst = max(frob_err(g, r["E"]) for g, r in zip(got, refs))
    assert worst <= 1e-9, worst
    # both device forms (workgroup-cooperative; the one-wave fit of the batched kernel) with their traces: the SIX parameters [r1; t1]
    # (src/spherical_estimator.cpp:140-144 leaves t1 free), the same number of Levenberg-Marquardt iterations, the same costs
    for wave in (False, True):
        E, x, it, status, c0, c1 = ransac.sampson_refine_probe_ex(gpu_ctx, u, v, lists, starts, inward=inward, wave=wave)
        assert max(frob_err(g, r["E"]) for g, r in zip(E, refs)) <= 1e-9
        assert max(np.abs(xx[:3] - r["x"][:3]).max() for xx, r in zip(x, refs)) <= 1e-9
        # t1 has a gauge direction (the scale of t) that only the damping holds: its component along t is rounding-sensitive, compare looser
        assert max(np.abs(xx[3:] - r["x"][3:]).max() for xx, r in zip(x, refs)) <= 1e-6
        assert [int(i) for i in it] == [r["iterations"] for r in refs] and (status == 0).all()
        assert max(abs(a - r["initial_cost"]) / r["initial_cost"] for a, r in zip(c0, refs)) <= 1e-9
        assert max(abs(a - r["final_cost"]) / r["final_cost"] for a, r in zip(c1, refs)) <= 1e-8
        assert max(np.abs(xx[3:] - [0, 0, 1.0 if inward else -1.0]).max() for xx in x) > 1e-5       # t1 moved on the device too


def test_least_squares_is_not_the_three_parameter_fit(gpu_ctx, oracle):
    """The negative test: rounds 1-2 pinned t1 (SURVEY a12's sentence); the reference does not (src/spherical_estimator.cpp:140-144).
    The device result must sit on the six-parameter minimum and AWAY from the three-parameter one by more than north_star's 1e-5."""
    from spherical_sfm_amd import ransac
    lists, starts, six, three = [], [], [], []
    pairs = []
    for seed in range(5):
        u, v, R, E, _ = synth.make_relative_pose_problem(500, seed=seed, noise=1 / 1000, rotation_deg=10)
        Rp = Rotation.from_rotvec(np.random.default_rng(seed).normal(size=3) * 0.01).as_matrix() @ R
        E0 = oracle.make_spherical_essential_matrix(Rp)
        s = np.arange(500, dtype=np.int32)
        a = oracle.sampson_least_squares_ex(u, v, s, E0); b = oracle.sampson_least_squares_ex(u, v, s, E0, r_only=True)
        _, x, it, status, c0, c1 = ransac.sampson_refine_probe_ex(gpu_ctx, u, v, [s], [E0], wave=bool(seed & 1))
        d6 = np.linalg.norm(x[0, :3] - a["x"][:3]); d3 = np.linalg.norm(x[0, :3] - b["x"][:3])
        assert d6 <= 1e-9 and d3 > 2e-5, (d6, d3)
        assert it[0] == a["iterations"] and c1[0] < b["final_cost"]


def test_decompose_probe_matches_oracle(gpu_ctx, oracle):
    """decompose_spherical_essential_matrix + so3exp (src/spherical_utils.cpp:16-66, spherical_estimator.cpp:159-164)."""
    from spherical_sfm_amd import ransac
    rng = np.random.default_rng(4)
    for inward in (False, True):
        Es, refs = [], []
        for k in range(200):
            ang = [1e-4, 0.02, 0.3, 1.0, 2.5, 3.1][k % 6]
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            E = oracle.make_spherical_essential_matrix(Rotation.from_rotvec(ax * ang).as_matrix(), inward)
            E = E * rng.uniform(0.2, 3.0) * (1 if k % 2 else -1) + (1e-7 * rng.normal(size=(3, 3)) if k % 5 == 0 else 0)
            Es.append(E); refs.append(oracle.decompose_spherical_essential_matrix(E, inward)[0])
        r, Rm = ransac.decompose_probe(gpu_ctx, Es, inward=inward)
        ang = [rot_err(Rotation.from_rotvec(a).as_matrix(), Rotation.from_rotvec(b).as_matrix()) for a, b in zip(r, refs)]
        assert max(ang) <= 1e-9, max(ang)
        assert max(np.abs(Rotation.from_rotvec(a).as_matrix() - M).max() for a, M in zip(r, Rm)) <= 1e-12


def test_nonminimal_solver_probe_matches_oracle(gpu_ctx, oracle):
    """SphericalEstimator::NonMinimalSolver (src/spherical_estimator.cpp:86-108) on samples of 4..9 rays."""
    from spherical_sfm_amd import ransac
    u, v, R, E, _ = synth.make_relative_pose_problem(60, seed=21, noise=5e-4)
    rng = np.random.default_rng(8)
    samples = [rng.choice(60, 4 + (k % 6), replace=False).astype(np.int32) for k in range(120)]
    ok, got = ransac.nonminimal_probe(gpu_ctx, u, v, samples)
    errs = []
    for s, k, g in zip(samples, ok, got):
        ko, ref = oracle.nonminimal_solver(u, v, s)
        assert k == ko == 1
        errs.append(frob_err(g, ref))
    errs = np.array(errs)
    # same elimination on both sides (pivoted QR, LU, companion roots); what is left is rounding times the conditioning of the 6x6 system
    assert np.median(errs) < 1e-12 and np.quantile(errs, 0.95) < 1e-8 and (errs > 1e-6).mean() <= 0.02, (np.median(errs), errs.max())


# ---- the whole control flow (row a13) ----------------------------------------------------------------------------------------
def _compare(out, k, o, u, v, oracle, tol=1e-9):
    same_trace = out["iterations"][k] == o["iterations"] and out["lo_runs"][k] == o["lo_runs"]
    same_mask = (out["inliers"][k] == o["inliers"]).all() and out["num_inliers"][k] == o["num_inliers"]
    close = frob_err(out["E"][k], o["E"]) <= tol and rot_err(out["R"][k], o["R"]) <= tol
    return same_trace, same_mask, close


def test_trace_mode_reproduces_the_oracle_pair_by_pair(gpu_ctx, oracle):
    """estimate_pairwise's options (num_lo_steps_ = 0, num_lsq_iterations_ = 0, final least squares; tools.cpp:314-318)."""
    from spherical_sfm_amd import ransac
    probs = _pairs(48, 150, 0.3, 1 / 600) + _pairs(16, 500, 0.45, 1 / 600, seed0=300)
    out = ransac.estimate_pairs(gpu_ctx, [(p[0], p[1]) for p in probs], THR, min_num_inliers=20)
    res = []
    for k, (u, v, R, E, inl) in enumerate(probs):
        o = oracle.lomsac_pair(u, v, THR, min_num_inliers=20)
        res.append(_compare(out, k, o, u, v, oracle))
        assert out["num_inliers"][k] == out["inliers"][k].sum()
        assert abs(out["scores"][k] - o["score"]) <= 1e-9 * o["score"] or not all(res[-1])
    res = np.array(res)
    # identical sample trace, identical inlier flags, E / R to 1e-9 on EVERY pair: with these options the only floating-point work between two
    # decisions is a minimal solve + a score, or one 6-parameter least-squares fit, and each of those agrees with the oracle to ~1e-10
    # (scripts/dev/trace_mismatch.py: 0 differing pairs of 256 on MI355X); the kernel has no atomics, so this does not vary from run to run
    assert res.all(), (res.mean(axis=0), np.nonzero(~res.all(1))[0])
    assert (out["iterations"] >= 100).all() and (out["lo_runs"] >= 1).all()


@pytest.mark.parametrize("poly", [False, True])
def test_trace_mode_with_local_optimization_steps(gpu_ctx, oracle, poly):
    """LORansacOptions defaults (num_lo_steps_ = 10, num_lsq_iterations_ = 4, ransac.h:62-88): NonMinimalSolver + iterated fits."""
    from spherical_sfm_amd import ransac
    probs = _pairs(24, 200, 0.35, 1 / 600, seed0=500)
    kw = dict(num_lo_steps=10, num_lsq_iterations=4, final_least_squares=0, min_num_inliers=20, use_poly_solver=int(poly))
    out = ransac.estimate_pairs(gpu_ctx, [(p[0], p[1]) for p in probs], THR, **kw)
    res = []
    for k, (u, v, R, E, inl) in enumerate(probs):
        o = oracle.lomsac_pair(u, v, THR, num_lo_steps=10, num_lsq_iterations=4, final_least_squares=False, min_num_inliers=20, use_poly=poly)
        res.append(_compare(out, k, o, u, v, oracle, tol=1e-8))
        assert rot_err(R, out["R"][k]) < 5e-3
    res = np.array(res)
    # Where a pair leaves the oracle's trace, find out WHY: replay the oracle's own call log on the device, call by call (same ray subset, same start
    # model), and name the first call whose output differs.  Measured on MI355X (scripts/dev/lsq_replay.py: 6528 LeastSquares calls): no
    # Levenberg-Marquardt stopping rule ever flips -- every fit has the oracle's iteration count and agrees to ~1e-9.  What differs is
    # NonMinimalSolver on an ill-conditioned 4..9-ray sample (the 6x6 elimination + companion roots amplify rounding to 1e-6 and more,
    # test_nonminimal_solver_probe_matches_oracle); the next GetInliers then admits a different ray and the runs part ways.
    for k in np.nonzero(~res.all(1))[0]:
        u, v = probs[k][0], probs[k][1]
        _, log = oracle.lsq_log(lambda: oracle.lomsac_pair(u, v, THR, num_lo_steps=10, num_lsq_iterations=4, final_least_squares=False, min_num_inliers=20, use_poly=poly))
        nm = oracle.lsq_log.nonminimal
        Eg, x, it, status, c0, c1 = ransac.sampson_refine_probe_ex(gpu_ctx, u, v, [l["sample"] for l in log], [l["E_in"] for l in log], wave=True)
        lsq_bad = [i for i, l in enumerate(log) if frob_err(Eg[i], l["E_out"]) > 1e-8 or it[i] != l["iterations"]]
        ok, En = ransac.nonminimal_probe(gpu_ctx, u, v, [m["sample"] for m in nm])
        nm_err = np.array([frob_err(e, m["E_out"]) for e, m in zip(En, nm)])
        print(f"pair {k}: {len(log)} LeastSquares calls, {len(lsq_bad)} differ (> 1e-8 or other iteration count); {len(nm)} NonMinimalSolver calls, "
              f"errors max {nm_err.max():.2e}, > 1e-9: {(nm_err > 1e-9).sum()}")
        assert not lsq_bad, (k, lsq_bad[:5])                        # never a flipped stopping rule
        assert (nm_err > 1e-10).any()                               # the divergence starts in a non-minimal solve
    assert res[:, 0].mean() >= 0.95 and res[:, 1].mean() >= 0.95 and res[:, 2].mean() >= 0.95, res.mean(axis=0)


def test_fast_shuffle_is_the_same_stream(gpu_ctx):
    from spherical_sfm_amd import ransac
    probs = _pairs(12, 400, 0.3, 1 / 600, seed0=700)
    kw = dict(num_lo_steps=3, num_lsq_iterations=2, min_num_inliers=20)
    a = ransac.estimate_pairs(gpu_ctx, [(p[0], p[1]) for p in probs], THR, fast_shuffle=1, **kw)
    b = ransac.estimate_pairs(gpu_ctx, [(p[0], p[1]) for p in probs], THR, fast_shuffle=0, **kw)
    assert (a["E"] == b["E"]).all() and (a["R"] == b["R"]).all() and (a["iterations"] == b["iterations"]).all()
    assert all((x == y).all() for x, y in zip(a["inliers"], b["inliers"]))


def test_ragged_tiny_and_large_pairs(gpu_ctx, oracle):
    """n < 3 (no model, ransac.h:137-141), n = 3 / 4 (ShuffleSample path, sampling.h:104-124), n = 3000 (> LDS capacity: rays from L2)."""
    from spherical_sfm_amd import ransac
    sizes = (0, 2, 3, 4, 5, 7, 50, 333, 3000)
    probs = [synth.make_relative_pose_problem(max(n, 1), seed=40 + n, noise=1 / 600, outlier_frac=0.2) for n in sizes]
    pairs = [(p[0][:n], p[1][:n]) for p, n in zip(probs, sizes)]
    out = ransac.estimate_pairs(gpu_ctx, pairs, THR, min_num_inliers=2)
    again = ransac.estimate_pairs(gpu_ctx, pairs, THR, min_num_inliers=2)
    assert (out["E"] == again["E"]).all() and (out["R"] == again["R"]).all()
    for k, n in enumerate(sizes):
        if n < 3:
            assert out["num_inliers"][k] == 0 and np.allclose(out["R"][k], np.eye(3)) and out["iterations"][k] == 0
            continue
        o = oracle.lomsac_pair(pairs[k][0], pairs[k][1], THR, min_num_inliers=2)
        tr, mk, cl = _compare(out, k, o, pairs[k][0], pairs[k][1], oracle, tol=1e-9)
        assert tr and mk and cl, (n, out["iterations"][k], o["iterations"], out["num_inliers"][k], o["num_inliers"])
    assert rot_err(probs[-1][2], out["R"][-1]) < 2e-3


def test_exhaustive_circle_streams_in_slabs(gpu_ctx, oracle, monkeypatch):
    """estimate_pairwise over every (i < j) of a 200-frame circle (BASELINE configs[3] in small): 19 900 pairs, 1.9 M rays, neighbours
    share hundreds of points and most far pairs none.  Default slabs vs forced tiny slabs: identical; a sample of pairs against the
    oracle; pairs below the acceptance threshold keep R = I (spherical_sfm_tools.cpp:410)."""
    from spherical_sfm_amd import ransac
    ptr, U, V, pairs, Rgt = synth.make_circle_pairs(200, 3000)
    assert len(pairs) == 19900
    out = ransac.estimate_flat(gpu_ctx, ptr, U, V, THR, min_num_inliers=20)
    monkeypatch.setenv("SSFM_RANSAC_SLAB_PAIRS", "1500"); monkeypatch.setenv("SSFM_RANSAC_SLAB_RAYS", "90000")
    small = ransac.estimate_flat(gpu_ctx, ptr, U, V, THR, min_num_inliers=20)
    monkeypatch.delenv("SSFM_RANSAC_SLAB_PAIRS"); monkeypatch.delenv("SSFM_RANSAC_SLAB_RAYS")
    for key in ("E", "R", "mask", "num_inliers", "scores", "iterations", "lo_runs"):
        assert (out[key] == small[key]).all(), key
    n = np.diff(ptr)
    accepted = out["num_inliers"] > 20
    assert (np.abs(out["R"][~accepted] - np.eye(3)) == 0).all() and (out["num_inliers"][n < 3] == 0).all()
    # neighbours are solved, to the accuracy the noise allows
    near = np.nonzero((pairs[:, 1] - pairs[:, 0] <= 10) & accepted)[0]
    assert len(near) >= 1900 and np.median([rot_err(Rgt[p], out["R"][p]) for p in near[::20]]) < 1e-3
    # a sample of 200 pairs (every size class) against the oracle
    rng = np.random.default_rng(0)
    sample = np.concatenate([rng.choice(np.nonzero(n >= 100)[0], 120, replace=False), rng.choice(np.nonzero((n >= 3) & (n < 100))[0], 80, replace=False)])
    ok = []
    for p in sample:
        u, v = U[ptr[p]:ptr[p + 1]], V[ptr[p]:ptr[p + 1]]
        o = oracle.lomsac_pair(u, v, THR, min_num_inliers=20)
        mask = out["mask"][ptr[p]:ptr[p + 1]].astype(bool)
        ok.append((out["iterations"][p] == o["iterations"] and out["lo_runs"][p] == o["lo_runs"], (mask == o["inliers"]).all(),
                   frob_err(out["E"][p], o["E"]) <= 1e-9 and rot_err(out["R"][p], o["R"]) <= 1e-9))
    ok = np.array(ok)
    assert ok[:, 0].mean() >= 0.97 and ok[:, 1].mean() >= 0.97 and ok[:, 2].mean() >= 0.97, ok.mean(axis=0)


def test_indexed_entry_point_equals_materialised_rays(gpu_ctx):
    """ssfm_ransac_batch_indexed (per-frame feature rays + per-pair match lists, rays gathered on the device) against ssfm_ransac_batch on the
    same ray pairs laid out by the host: every output bit for bit (same kernels, same random streams), several slabs, an empty pair, and an
    out-of-range feature index refused."""
    import os
    from spherical_sfm_amd import ransac, _lib
    rng = np.random.default_rng(11)
    nframes = 9
    probs = [synth.make_relative_pose_problem(120 + 7 * k, seed=300 + k, noise=1e-3, outlier_frac=0.25, rotation_deg=3 + k) for k in range(nframes - 1)]
    # frame f holds the u-rays of problem f (as its features, shuffled) and the v-rays of problem f-1 behind them
    feat = [[] for _ in range(nframes)]; where_u = []; where_v = []
    for k, (u, v, *_rest) in enumerate(probs):
        pu = rng.permutation(len(u)); pv = rng.permutation(len(v))
        base_u = len(feat[k]); feat[k].extend(u[pu]); iu = np.empty(len(u), np.int32); iu[pu] = base_u + np.arange(len(u)); where_u.append(iu)
        base_v = len(feat[k + 1]); feat[k + 1].extend(v[pv]); iv = np.empty(len(v), np.int32); iv[pv] = base_v + np.arange(len(v)); where_v.append(iv)
    feat_ptr = np.zeros(nframes + 1, np.int32)
    for f in range(nframes): feat_ptr[f + 1] = feat_ptr[f] + len(feat[f])
    rays = np.concatenate([np.asarray(f).reshape(-1, 3) for f in feat])
    f0 = []; f1 = []; mp = [0]; m0 = []; m1 = []; U = []; V = []
    for rep in range(3):                                   # 24 pairs + an empty one
        for k, (u, v, *_rest) in enumerate(probs):
            f0.append(k); f1.append(k + 1); m0.append(where_u[k]); m1.append(where_v[k]); mp.append(mp[-1] + len(u)); U.append(u); V.append(v)
        if rep == 1: f0.append(2); f1.append(5); mp.append(mp[-1])
    m0 = np.concatenate(m0); m1 = np.concatenate(m1); U = np.concatenate(U); V = np.concatenate(V); mp = np.array(mp, np.int32)
    thr = (2e-3) ** 2
    old = os.environ.get("SSFM_RANSAC_SLAB_PAIRS"); os.environ["SSFM_RANSAC_SLAB_PAIRS"] = "7"      # several slabs through the double buffer
    try:
        a = ransac.estimate_flat(gpu_ctx, mp, U, V, thr, min_num_inliers=20)
        b = ransac.estimate_indexed(gpu_ctx, feat_ptr, rays, f0, f1, mp, m0, m1, thr, min_num_inliers=20)
    finally:
        if old is None: del os.environ["SSFM_RANSAC_SLAB_PAIRS"]
        else: os.environ["SSFM_RANSAC_SLAB_PAIRS"] = old
    for key in ("E", "R", "mask", "num_inliers", "scores", "iterations", "lo_runs"):
        assert np.array_equal(a[key], b[key]), key
    assert (a["num_inliers"] > 20).sum() >= 20
    bad = m0.copy(); bad[5] = 10 ** 6
    with pytest.raises(_lib.SsfmError):
        ransac.estimate_indexed(gpu_ctx, feat_ptr, rays, f0, f1, mp, bad, m1, thr)


def test_configs3_full_size_properties(gpu_ctx, oracle):
    """BASELINE configs[3] at its full size on one GPU: the 1 999 000 image pairs of a 2000-frame exhaustive circle x 500 correspondences (SURVEY 8d:
    30 % outliers, noise 1 px / f, threshold (2 px / f)^2, f = 1000) through ssfm_ransac_batch_indexed -- per-frame feature rays once, 8 bytes of match
    indices per correspondence.  The pair list cycles through 2000 distinct relative-pose problems (frame k = problem k), which gives the
    size-independent properties: every pair is accepted and its rotation sits within the noise of the generating one; equal inputs give
    bit-equal outputs wherever they sit in the stream of slabs; 200 sampled pairs equal the oracle's run (same trace)."""
    import time
    from spherical_sfm_amd import ransac
    F = 1000.0; thr = (2 / F) ** 2; NC = 500; POOL = 2000; TOTAL = 1999000; PER = 100000
    probs = [synth.make_relative_pose_problem(NC, seed=1000 + k, noise=1 / F, outlier_frac=0.3, rotation_deg=1 + (k % 60)) for k in range(POOL)]
    feat_ptr = (np.arange(POOL + 1, dtype=np.int64) * 2 * NC).astype(np.int32)
    feat_rays = np.ascontiguousarray(np.concatenate([np.concatenate([p[0], p[1]]) for p in probs]))
    ptr = (np.arange(PER + 1, dtype=np.int64) * NC).astype(np.int32)
    m0 = np.tile(np.arange(NC, dtype=np.int32), PER); m1 = m0 + NC
    first = None; done = 0; t0 = time.perf_counter(); accepted = 0
    while done < TOTAL:
        n = min(PER, TOTAL - done)
        fr = ((done + np.arange(n)) % POOL).astype(np.int32)
        o = ransac.estimate_indexed(gpu_ctx, feat_ptr, feat_rays, fr, fr, ptr[:n + 1], m0[:n * NC], m1[:n * NC], thr, min_num_inliers=20)
        accepted += int((o["num_inliers"] > 20).sum())
        if first is None:
            first = {k: o[k][:POOL].copy() for k in ("E", "R", "num_inliers", "scores", "iterations", "lo_runs")}
            first["mask"] = o["mask"][:POOL * NC].copy()
        # determinism across the whole stream: pair p repeats problem p mod 2000 with the same random streams
        for key in ("E", "R", "num_inliers", "scores", "iterations", "lo_runs"):
            assert np.array_equal(o[key], first[key][fr]), (key, done)
        if done == 0 or done + n == TOTAL:
            assert np.array_equal(o["mask"].reshape(n, NC), first["mask"].reshape(POOL, NC)[fr])
        done += n
    dt = time.perf_counter() - t0
    assert accepted == TOTAL
    errs = np.array([rot_err(probs[k][2], first["R"][k]) for k in range(POOL)])
    assert errs.max() < 5e-3 and np.median(errs) < 5e-4, (errs.max(), np.median(errs))
    inl_gt = np.array([p[4] for p in probs])
    agree = (first["mask"].reshape(POOL, NC).astype(bool) == inl_gt).mean()
    assert agree > 0.95, agree             # the 2 px threshold at 1 px noise cuts the tail of the true inliers (0.966 measured)
    for k in range(0, POOL, 10):                                   # 200 pairs against the oracle: same trace, same flags, same rotation
        r = oracle.lomsac_pair(probs[k][0], probs[k][1], thr, min_num_inliers=20)
        assert r["iterations"] == first["iterations"][k] and r["lo_runs"] == first["lo_runs"][k] and r["num_inliers"] == first["num_inliers"][k], k
        assert np.array_equal(r["inliers"], first["mask"].reshape(POOL, NC)[k].astype(bool)) and rot_err(r["R"], first["R"][k]) <= 1e-9, k
    print(f"configs[3]: {TOTAL} pairs in {dt:.2f} s ({TOTAL / dt:.3e} pairs/s incl. the checks)")
