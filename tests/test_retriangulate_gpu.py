"""GPU Retriangulate (ssfm_retriangulate through the C ABI) against the oracle, which replays the reference's LO-MSAC
(src/sfm.cpp:156-192, src/triangulation_estimator.cpp:46-127, include/RansacLib/ransac.h) with its std::mt19937 streams.

The default device mode replays the same trace (retriangulate.hip): the sampler's pair sequence and the local optimisation's random words come
from libstdc++'s generators on the host, the device walks RansacLib's control flow draw for draw.  RANSAC revisits the same observation pair many
times and starts a local optimisation whenever a score is lower than the best by ANY margin, so the replay only works if the minimal solver and the
scoring round every operation like the CPU: both sides are compiled without fused multiply-adds and sum in one documented order, and these tests
compare the pieces for EQUALITY (==), then the whole run: identical iteration counts, local-optimisation runs, inlier sets, points <= 1e-9.
The enumerating kernel of rounds 1-2 (ssfm_retriangulate_mode, SSFM_RETRI_MODE_ENUMERATE) keeps its statistical comparison at the end of the file."""
import dataclasses

import numpy as np
import pytest

from spherical_sfm_amd import ba, synth

pytestmark = pytest.mark.gpu


def _msac_scores(prob, X):
    R = synth.so3exp(prob.cameras[:, 3:]); t = prob.cameras[:, :3]
    pc = np.einsum('nij,nj->ni', R[prob.obs_cam], X[prob.obs_pt]) + t[prob.obs_cam]
    with np.errstate(divide="ignore", invalid="ignore"):
        e = ((prob.focal * pc[:, :2] / pc[:, 2:3] - prob.obs_xy) ** 2).sum(1)
    e = np.where(pc[:, 2] < 0, np.inf, e)
    return np.bincount(prob.obs_pt, np.minimum(e, 4.0), len(X))


def _compare(prob, oracle, agree_frac=None):
    """the whole trace: same RansacStatistics, same inlier sets, same zeroing decisions, points to 1e-9"""
    Xo, no, ito, loo, flo = oracle.retriangulate_ex(prob, 16)
    ctx = ba.Context()
    Xg, ng, itg, log_, flg = ba.retriangulate_ex(ctx, prob)
    ctx.close()
    bad = np.nonzero((itg != ito) | (log_ != loo) | (ng != no))[0]
    assert len(bad) == 0, (len(bad), bad[:10], itg[bad[:10]], ito[bad[:10]], log_[bad[:10]], loo[bad[:10]])
    assert np.array_equal(flg, flo)                                    # identical inlier sets, observation by observation
    zg, zo = ~Xg.any(1), ~Xo.any(1)
    assert np.array_equal(zg, zo)
    rel = np.linalg.norm(Xg - Xo, axis=1)[~zo] / np.linalg.norm(Xo[~zo], axis=1)
    assert rel.size == 0 or rel.max() <= 1e-9, (rel.max(), (rel > 1e-9).sum())
    return Xg, ng, Xo, no


def _small_problem(seed=3, Nc=60, Np=400, K=6, corrupt=0.1):
    prob = synth.make_circle(Nc, Np, K, rot_noise_deg=0.0, pixel_noise=0.5, seed=seed)
    if corrupt:
        synth.corrupt_observations(prob, corrupt, seed=5)
    return prob


# ---- the estimator's pieces, bit for bit -----------------------------------------------------------------------------------------
def test_dlt_scores_and_least_squares_are_bit_identical(oracle, gpu_ctx):
    """NonMinimalSolver (2..6 observations, both orders of a pair), ScoreModel / GetInliers, LeastSquares: the device rounds like the CPU."""
    prob = _small_problem()
    rng = np.random.default_rng(0)
    Np, K = 400, 6
    # DLT: every ordered pair of a few points + random larger samples
    tp, lists = [], []
    for j in range(0, Np, 7):
        for a in range(K):
            for b in range(K):
                if a != b:
                    tp.append(j); lists.append([a, b])
        for m in (3, 4, 5, 6):
            tp.append(j); lists.append(list(rng.permutation(K)[:m]))
        tp.append(j); lists.append([2, 2]); tp.append(j); lists.append([1, 0, 0, 0])        # degenerate samples (zero-padded short base sets)
    g = ba.tri_probe(gpu_ctx, prob, 0, tp, lists); o = oracle.tri_probe(prob, 0, tp, lists)
    same = (g == o) | (np.isnan(g) & np.isnan(o))
    assert same.all(), (np.nonzero(~same.all(1))[0][:10], g[~same.all(1)][:3], o[~same.all(1)][:3])
    # the two orders of a pair are NOT bit-identical in general -- that is why the replay needs exact arithmetic
    Xs = g[:, :3]
    # scores / inlier counts / single errors of those models
    g2 = ba.tri_probe(gpu_ctx, prob, 2, tp, lists, Xs); o2 = oracle.tri_probe(prob, 2, tp, lists, Xs)
    same2 = (g2 == o2) | (np.isnan(g2) & np.isnan(o2))
    assert same2.all(), (g2[~same2.all(1)][:3], o2[~same2.all(1)][:3])
    # least squares from the DLT points on subsets (3..6 observations) and on everything
    ok = np.isfinite(Xs).all(1)
    tp3 = [t for t, k in zip(tp, ok) if k][:600]; X3 = Xs[ok][:600]
    l3 = [sorted(rng.permutation(K)[:rng.integers(2, K + 1)]) for _ in tp3]
    g3 = ba.tri_probe(gpu_ctx, prob, 1, tp3, l3, X3); o3 = oracle.tri_probe(prob, 1, tp3, l3, X3)
    assert np.array_equal(g3[:, 3], o3[:, 3])                           # the same number of Levenberg-Marquardt iterations
    d = np.abs(g3[:, :3] - o3[:, :3]).max(1) / np.abs(o3[:, :3]).max(1)
    # pow() in the radius update is the one operation that is not IEEE-exact on both sides; it touches the last bits of a damping term
    assert (d == 0).mean() >= 0.98 and d.max() <= 1e-12, ((d == 0).mean(), d.max())


def test_swapped_pairs_really_differ_in_the_last_bits(oracle):
    """Why the arithmetic has to be exact: the two orders of an observation pair give scores that differ by rounding only."""
    prob = _small_problem()
    tp = [j for j in range(50) for _ in range(2)]; lists = [[0, 3], [3, 0]] * 50
    o = oracle.tri_probe(prob, 0, tp, lists)
    sc = oracle.tri_probe(prob, 2, tp, lists, o[:, :3])[:, 0].reshape(-1, 2)
    diff = np.abs(sc[:, 0] - sc[:, 1])
    assert (diff > 0).any() and (diff <= 1e-9 * np.abs(sc).max(1)).all()


def test_noise_free_exact(oracle, gpu_ctx):
    prob = synth.make_circle(60, 3000, 6, rot_noise_deg=0.0, pixel_noise=0.0, seed=11)
    prob = dataclasses.replace(prob, cameras=prob.gt_cameras.copy(), points=np.zeros_like(prob.points))
    Xg, ng = ba.retriangulate(gpu_ctx, prob)
    assert (ng == 6).all()
    assert (np.linalg.norm(Xg - prob.gt_points, axis=1) / np.linalg.norm(prob.gt_points, axis=1)).max() < 1e-7
    Xo, no = oracle.retriangulate(prob, 16)
    assert (np.linalg.norm(Xg - Xo, axis=1) / np.linalg.norm(Xo, axis=1)).max() < 1e-7 and (no == ng).all()


def test_short_tracks_and_inconsistent_tracks_become_zero(oracle, gpu_ctx):
    prob = synth.make_circle(60, 600, 6, rot_noise_deg=0.0, pixel_noise=0.0, seed=12)
    prob = dataclasses.replace(prob, cameras=prob.gt_cameras.copy())
    keep = ~((prob.obs_pt < 100) & (np.arange(len(prob.obs_pt)) % 6 >= 2))       # 2 observations left
    keep &= ~((prob.obs_pt >= 100) & (prob.obs_pt < 200) & (np.arange(len(prob.obs_pt)) % 6 >= 3))   # 3 left ...
    prob = dataclasses.replace(prob, obs_xy=prob.obs_xy[keep].copy(), obs_cam=prob.obs_cam[keep], obs_pt=prob.obs_pt[keep])
    first = np.nonzero((prob.obs_pt >= 100) & (prob.obs_pt < 150))[0][::3]
    prob.obs_xy[first] += [0.0, 70.0]                                               # ... one of them wrong for 100..149
    Xg, ng = ba.retriangulate(gpu_ctx, prob)
    Xo, no = oracle.retriangulate(prob, 16)
    assert not Xg[:150].any() and not Xo[:150].any()
    assert Xg[150:].any(axis=1).all() and np.array_equal(ng[150:], no[150:])
    assert (np.linalg.norm(Xg - Xo, axis=1)[150:] / np.linalg.norm(Xo[150:], axis=1)).max() < 1e-6


@pytest.mark.parametrize("Nc,Np,K", [(60, 2000, 6), (120, 6000, 10), (60, 1500, 3), (90, 1500, 4), (200, 800, 25)])
def test_noisy_tracks_with_outliers_replay_the_oracle_trace(oracle, Nc, Np, K):
    prob = synth.make_circle(Nc, Np, K, rot_noise_deg=0.0, pixel_noise=0.5, seed=3)
    bad = synth.corrupt_observations(prob, 0.1, seed=5)
    Xg, ng, Xo, no = _compare(prob, oracle)
    mask = np.zeros(Np, bool); mask[bad] = True
    assert (ng[mask] <= K - 1).all()                               # the displaced observation is never an inlier


def test_ragged_tracks_replay_the_oracle_trace(oracle):
    """track lengths 2..14 in one call: one sampler sequence per distinct length (n = 3 takes the shuffle branch of UniformSampling)"""
    prob = synth.make_circle(120, 3000, 14, rot_noise_deg=0.0, pixel_noise=0.7, seed=8)
    synth.corrupt_observations(prob, 0.15, seed=6)
    rng = np.random.default_rng(1)
    keep_n = rng.integers(2, 15, 3000)
    rank = np.zeros(len(prob.obs_pt), int)
    order = np.argsort(prob.obs_pt, kind="stable"); start = np.searchsorted(prob.obs_pt[order], np.arange(3000))
    rank[order] = np.arange(len(order)) - start[prob.obs_pt[order]]
    sel = rank < keep_n[prob.obs_pt]                                    # the first keep_n observations of every point survive
    prob = dataclasses.replace(prob, obs_xy=prob.obs_xy[sel].copy(), obs_cam=prob.obs_cam[sel], obs_pt=prob.obs_pt[sel])
    Xg, ng, Xo, no = _compare(prob, oracle)
    assert not Xg[keep_n == 2].any()


@pytest.mark.parametrize("case", range(6))
def test_random_shapes_noise_levels_and_flipped_cameras(oracle, case):
    """scripts/soak_retri_trace.py in small (60 cases / 94 563 points there: no difference at all, points bit-identical): heavy noise, up to 60 % of the
    points with a gross outlier, ragged tracks, cameras turned around so that points fall behind them (DBL_MAX errors)."""
    rng = np.random.default_rng(700 + case)
    Nc = int(rng.choice([24, 60, 200])); K = min(int(rng.choice([3, 5, 8, 12, 20])), Nc // 4); Np = int(rng.integers(300, 1500))
    prob = synth.make_circle(Nc, Np, K, rot_noise_deg=0.5, pixel_noise=float(rng.choice([0.2, 1.5, 4.0])), seed=int(rng.integers(1, 10 ** 6)), check_in_frame=False, xy_range=0.25)
    synth.corrupt_observations(prob, float(rng.choice([0.1, 0.3, 0.6])), seed=case)
    if case % 2 == 0:
        cams = prob.cameras.copy(); cams[::7, 3:] += [0.0, np.pi, 0.0]; prob = dataclasses.replace(prob, cameras=cams)
    _compare(prob, oracle)


def test_degenerate_inputs(oracle, gpu_ctx):
    """no observations at all; one point; only short tracks; a camera index out of range is ignored like a missing camera (src/sfm.cpp:166-167)"""
    base = synth.make_circle(24, 40, 4, rot_noise_deg=0.0, pixel_noise=0.2, seed=5, check_in_frame=False, xy_range=0.25)
    empty = dataclasses.replace(base, obs_xy=np.zeros((0, 2)), obs_cam=np.zeros(0, np.int32), obs_pt=np.zeros(0, np.int32))
    X, n, it, lo, fl = ba.retriangulate_ex(gpu_ctx, empty)
    assert not X.any() and not n.any() and not it.any() and len(fl) == 0
    one = dataclasses.replace(base, points=base.points[:1].copy(), pt_fixed=base.pt_fixed[:1], obs_xy=base.obs_xy[:4].copy(), obs_cam=base.obs_cam[:4], obs_pt=base.obs_pt[:4],
                              gt_points=base.gt_points[:1])
    Xg, ng, itg, log_, flg = ba.retriangulate_ex(gpu_ctx, one); Xo, no, ito, loo, flo = oracle.retriangulate_ex(one, 1)
    assert np.array_equal(Xg, Xo) and np.array_equal(ng, no) and np.array_equal(itg, ito) and np.array_equal(flg, flo) and ng[0] == 4
    short = dataclasses.replace(base, obs_xy=base.obs_xy[base.obs_pt % 2 == 0][::2].copy(), obs_cam=base.obs_cam[base.obs_pt % 2 == 0][::2], obs_pt=base.obs_pt[base.obs_pt % 2 == 0][::2])
    Xs, ns = ba.retriangulate(gpu_ctx, short)
    assert not Xs.any() and not ns.any()                             # two observations per point at most: everything is zeroed (src/sfm.cpp:173)
    bad = dataclasses.replace(base, obs_cam=np.where(np.arange(len(base.obs_cam)) % 4 == 3, 99, base.obs_cam).astype(np.int32))
    Xb, nb, itb, lob, flb = ba.retriangulate_ex(gpu_ctx, bad)
    assert (nb <= 3).all() and not flb[np.arange(len(bad.obs_cam)) % 4 == 3].any() and Xb.any(1).sum() > 30


def test_full_size_config2(oracle):
    """BASELINE config 2 sizes: 300 cameras x 100k points x 600k observations -- 100 % identical inlier sets, points <= 1e-9."""
    prob = synth.make_circle(300, 100000, 6, rot_noise_deg=0.0, pixel_noise=0.5, seed=3)
    synth.corrupt_observations(prob, 0.1, seed=5)
    _compare(prob, oracle)


def test_small_random_stream_table_is_extended(oracle, gpu_ctx, monkeypatch):
    """a lane that runs off the uploaded raw words makes the host repeat the launch with a longer table: same answer"""
    prob = _small_problem(Np=300)
    X0, n0 = ba.retriangulate(gpu_ctx, prob)
    monkeypatch.setenv("SSFM_RETRI_WORDS", "64")
    X1, n1 = ba.retriangulate(gpu_ctx, prob)
    assert np.array_equal(X0, X1) and np.array_equal(n0, n1)


def test_enumerating_mode_agrees_statistically(oracle, gpu_ctx):
    """ssfm_retriangulate_mode(SSFM_RETRI_MODE_ENUMERATE): the kernel of rounds 1-2 (every pair once, no random stream) -- a different result from the trace
    replay (asserted: otherwise this test would be running the wrong kernel), statistically the same reconstruction"""
    prob = synth.make_circle(60, 2000, 6, rot_noise_deg=0.0, pixel_noise=0.5, seed=3)
    synth.corrupt_observations(prob, 0.1, seed=5)
    Xo, no = oracle.retriangulate(prob, 16)
    Xg, ng = ba.retriangulate(gpu_ctx, prob, mode=ba.RETRI_MODE_ENUMERATE)
    Xt, _ = ba.retriangulate(gpu_ctx, prob, mode=ba.RETRI_MODE_TRACE)
    assert not np.array_equal(Xg, Xt) and np.array_equal(Xt, ba.retriangulate(gpu_ctx, prob)[0])          # the default is the trace replay
    zg, zo = ~Xg.any(1), ~Xo.any(1)
    assert (zg != zo).mean() <= 1e-3 and (ng == no).mean() >= 0.99
    both = ~zg & ~zo & (ng == no)
    rel = np.linalg.norm(Xg - Xo, axis=1)[both] / np.linalg.norm(Xo[both], axis=1)
    assert np.median(rel) <= 1e-5 and np.quantile(rel, 0.999) <= 2e-4
    sg, so = _msac_scores(prob, Xg)[~zg & ~zo].sum(), _msac_scores(prob, Xo)[~zg & ~zo].sum()
    assert abs(sg - so) <= 2e-4 * so


def test_duplicate_and_unsorted_observations(oracle, gpu_ctx):
    """observation order does not matter; a repeated (camera, point) key keeps its last value (SparseMatrix semantics)"""
    prob = synth.make_circle(60, 500, 6, rot_noise_deg=0.0, pixel_noise=0.3, seed=13)
    X0, n0 = ba.retriangulate(gpu_ctx, prob)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(prob.obs_pt))
    dup_xy = np.concatenate([prob.obs_xy[:200] + 500.0, prob.obs_xy])[np.concatenate([np.arange(200), 200 + perm])]
    dup_c = np.concatenate([prob.obs_cam[:200], prob.obs_cam[perm]]); dup_p = np.concatenate([prob.obs_pt[:200], prob.obs_pt[perm]])
    p2 = dataclasses.replace(prob, obs_xy=dup_xy, obs_cam=dup_c, obs_pt=dup_p)
    X1, n1 = ba.retriangulate(gpu_ctx, p2)
    assert np.array_equal(n0, n1) and np.abs(X0 - X1).max() <= 1e-9 * np.abs(X0).max()


def test_gpu_against_retriangulate_golden(gpu_ctx):
    """HIP path vs the committed fixture tests/golden/retriangulate.npz (no oracle call)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "retriangulate.npz"))
    Np = len(g["points"])
    p = synth.BAProblem(cameras=g["cameras"], points=np.ones((Np, 3)), focal=float(g["focal"]), obs_xy=g["obs_xy"], obs_cam=g["obs_cam"], obs_pt=g["obs_pt"],
                        rot_fixed=np.zeros(len(g["cameras"]), np.uint8), trans_fixed=np.ones(len(g["cameras"]), np.uint8), pt_fixed=np.zeros(Np, np.uint8),
                        focal_fixed=True, gt_cameras=g["cameras"], gt_points=g["points"], gt_focal=0.0)
    X, nin, it, lo, fl = ba.retriangulate_ex(gpu_ctx, p)
    assert np.array_equal(it, g["iterations"]) and np.array_equal(lo, g["lo_runs"]) and np.array_equal(fl, g["inlier_flags"])
    assert np.array_equal(nin, g["num_inliers"])
    nz = g["points"].any(1)
    assert np.array_equal(X.any(1), nz)
    rel = np.linalg.norm(X - g["points"], axis=1)[nz] / np.linalg.norm(g["points"][nz], axis=1)
    assert rel.max() <= 1e-9
