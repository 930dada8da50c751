"""Signature groups (k_schur_gram, ba_kernels.h): points that share their camera list take the Gram/MFMA path, the others the pair lists and
k_cam_sums2 -- and a problem usually has both.  Mixed problems against the oracle and against the same solve with grouping switched off."""
import dataclasses
import numpy as np
import pytest

from spherical_sfm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _keep_every_group(monkeypatch):
    """These tests are about the kernels behind the signature groups: the planner's cost model (round 4: it drops groups whose launches would cost more than the
    pair lists) is switched off so that small mixed problems still take them."""
    monkeypatch.setenv("SSFM_GRAM_MODEL", "0")


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def mixed_problem(seed, K, spherical, focal_fixed, loose_frac=0.3, per_cam=70):
    """Runs of ~per_cam points per camera window (grouped), of which loose_frac lose one observation (their camera list becomes their own:
    pair lists), a few constant points inside the groups, extra constant cameras, one camera seen only by grouped points."""
    rng = np.random.default_rng(seed)
    Nc = max(24, int(np.ceil((K - 1) * 360.0 / 45.0)) + 2) + int(rng.integers(0, 20))
    Np = per_cam * Nc
    p = synth.make_circle(Nc, Np, K, spherical=spherical, focal_fixed=focal_fixed, check_in_frame=False, seed=seed, xy_range=0.25)
    keep = np.ones(len(p.obs_cam), bool)
    if K > 2 and loose_frac > 0:
        # loose points come in blocks, so that whole runs survive between them (a run needs >= 32 identical neighbours)
        blocks = rng.random(Np // 16 + 1) < loose_frac
        loose = np.nonzero(blocks[np.arange(Np) // 16])[0]
        drop = loose * K + rng.integers(0, K, size=len(loose))          # observation j of point q sits at q K + j (make_circle is point-major)
        keep[drop] = False
    pt_fixed = p.pt_fixed.copy(); pt_fixed[rng.choice(Np, size=Np // 40 + 1, replace=False)] = 1
    rot_fixed = p.rot_fixed.copy(); rot_fixed[rng.choice(Nc, size=2, replace=False)] = 1
    trans_fixed = p.trans_fixed.copy()
    if not spherical:
        trans_fixed[rng.choice(Nc, size=3, replace=False)] = 1
    return dataclasses.replace(p, obs_xy=p.obs_xy[keep], obs_cam=p.obs_cam[keep], obs_pt=p.obs_pt[keep], pt_fixed=pt_fixed, rot_fixed=rot_fixed,
                               trans_fixed=trans_fixed)


CASES = [(2, True, True), (3, False, True), (4, True, False), (5, False, False), (6, True, True), (6, False, True), (7, False, False), (8, True, False), (8, False, True)]


@pytest.mark.parametrize("K,spherical,focal_fixed", CASES)
def test_mixed_groups_match_oracle_and_ungrouped_solve(gpu_ctx, oracle, monkeypatch, K, spherical, focal_fixed):
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    monkeypatch.setenv("SSFM_GRAM_KMIN", "2")
    monkeypatch.setenv("SSFM_GRAM_BACKSUB", "1" if K % 2 == 0 else "0")  # default: k_gram_backsub from 7 observations per point on average
    p = mixed_problem(500 + K, K, spherical, focal_fixed)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] and abs(s["iterations"] - os_["iterations"]) <= 1 and s["pcg_iterations_total"] == 0
    assert abs(s["final_cost"] - os_["final_cost"]) <= 1e-7 * os_["final_cost"]
    assert rel_err(cams, ocams) <= 1e-5 and abs(f - of) <= 1e-5 * of
    used = np.linalg.norm(opts, axis=1) > 0
    assert (np.linalg.norm(pts[used] - opts[used], axis=1) / np.linalg.norm(opts[used], axis=1)).max() <= 1e-4
    monkeypatch.setenv("SSFM_GRAM", "0")
    c0, p0, f0, s0 = ba.optimize(gpu_ctx, p)
    assert s0["iterations"] == s["iterations"] and rel_err(cams, c0) <= 1e-7 and rel_err(pts, p0) <= 1e-7 and abs(f - f0) <= 1e-9 * f0
    assert abs(s0["final_cost"] - s["final_cost"]) <= 1e-10 * s["final_cost"]


@pytest.mark.parametrize("loss", [0, 1, 2])
def test_groups_with_every_loss(gpu_ctx, oracle, monkeypatch, loss):
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = mixed_problem(77, 6, False, False)
    kw = dict(loss_type=loss, loss_scale=2.0)
    cams, pts, f, s = ba.optimize(gpu_ctx, p, **kw)
    ocams, opts, of, os_ = oracle.ba_solve(p, **kw)
    assert abs(s["iterations"] - os_["iterations"]) <= 1 and abs(s["final_cost"] - os_["final_cost"]) <= 1e-7 * os_["final_cost"]
    assert rel_err(cams, ocams) <= 1e-5 and abs(f - of) <= 1e-5 * of


def test_which_kernels_run(gpu_ctx, monkeypatch):
    """All points grouped: neither the pair kernel nor k_cam_sums2 runs; mixed: all three; grouping off or K > 8: no Gram kernel."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")

    def kernels(p):
        adj = ba.BundleAdjuster(gpu_ctx, p); adj.set_profiling(True); adj.run()
        kt = adj.kernel_times(); adj.close()
        return {n for n in kt if kt[n]["launches"] > 0}

    full = synth.make_circle(60, 60 * 70, 6, spherical=False, focal_fixed=True)
    k = kernels(full)
    assert "k_schur_gram" in k and "k_schur_pairs2" not in k and "k_cam_sums2" not in k
    assert "k_gram_backsub" in k and "k_point_backsub" not in k          # 6 observations per point (round 6, k_gram_backsub2): the grouped back substitution
    k = kernels(synth.make_circle(60, 60 * 70, 5, spherical=False, focal_fixed=True))
    assert "k_point_backsub" in k and "k_gram_backsub" not in k          # 5 per point: the lane-per-point back substitution
    full8 = synth.make_circle(80, 80 * 70, 8, spherical=False, focal_fixed=True, check_in_frame=False, xy_range=0.25)
    k = kernels(full8)
    assert "k_gram_backsub" in k and "k_point_backsub" not in k          # 8 per point, every point grouped: the residual check rides with k_gram_backsub
    k = kernels(mixed_problem(9, 8, False, True))
    assert {"k_schur_gram", "k_schur_pairs2", "k_cam_sums2", "k_gram_backsub", "k_point_backsub"} <= k
    k = kernels(mixed_problem(9, 6, False, True))
    assert {"k_schur_gram", "k_schur_pairs2", "k_cam_sums2"} <= k
    monkeypatch.setenv("SSFM_GRAM_KMIN", "4")
    k = kernels(synth.make_circle(60, 60 * 70, 3, spherical=False, focal_fixed=True))
    assert "k_schur_gram" not in k                                       # SSFM_GRAM_KMIN: shorter camera lists stay with the pair lists
    monkeypatch.delenv("SSFM_GRAM_KMIN")
    k = kernels(synth.make_circle(60, 60 * 70, 3, spherical=False, focal_fixed=True))
    assert "k_schur_gram" in k and "k_schur_pairs2" not in k             # 18 Gram rows: one 16-row tile + the 4x4x4 tail
    monkeypatch.setenv("SSFM_GRAM", "0")
    k = kernels(full)
    assert "k_schur_gram" not in k and {"k_schur_pairs2", "k_cam_sums2"} <= k
    monkeypatch.delenv("SSFM_GRAM")
    wide = synth.make_circle(120, 120 * 40, 10, spherical=True, focal_fixed=True, check_in_frame=False, xy_range=0.2)
    k = kernels(wide)
    assert "k_schur_gram" not in k and "k_schur_pairs2" in k


@pytest.mark.parametrize("pts_per_task", [8, 24, 40, 200])
def test_task_length_does_not_change_the_answer(gpu_ctx, monkeypatch, pts_per_task):
    """Tasks that end inside a sub-chunk of 8 points, single-sub-chunk tasks, tasks longer than a run."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = mixed_problem(31, 7, False, False, per_cam=45)
    monkeypatch.setenv("SSFM_GRAM_BACKSUB", "1")
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    monkeypatch.setenv("SSFM_GRAM_PTS", str(pts_per_task))
    c1, p1, f1, s1 = ba.optimize(gpu_ctx, p)
    assert s1["iterations"] == s["iterations"] and rel_err(cams, c1) <= 1e-8 and rel_err(pts, p1) <= 1e-8 and abs(f - f1) <= 1e-10 * f


@pytest.mark.parametrize("split", [1, 2, 3, 5, 8])
def test_waves_per_task_of_the_grouped_back_substitution(gpu_ctx, monkeypatch, split):
    """k_gram_backsub2 with several waves per task (SSFM_GBS_SPLIT): shares that end inside a sub-chunk, empty shares (more waves than sub-chunks), every share count
    against the lane-per-point kernel."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = mixed_problem(17, 7, False, False, per_cam=45)
    monkeypatch.setenv("SSFM_GRAM_BACKSUB", "0")
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    monkeypatch.setenv("SSFM_GRAM_BACKSUB", "1")
    monkeypatch.setenv("SSFM_GRAM_PTS", "40")                           # five sub-chunks per task
    monkeypatch.setenv("SSFM_GBS_SPLIT", str(split))
    c1, p1, f1, s1 = ba.optimize(gpu_ctx, p)
    assert s1["iterations"] == s["iterations"] and rel_err(cams, c1) <= 1e-8 and rel_err(pts, p1) <= 1e-8 and abs(f - f1) <= 1e-10 * f


def multi_k_problem(seed, spherical, focal_fixed, block=48):
    """An 8-observation circle whose points keep only their first k cameras, k drawn per block of `block` consecutive points from 3..8: runs of every length class
    (one, two and three 16-row tiles) in ONE problem, plus blocks shorter than a run (16 points: pair lists)."""
    rng = np.random.default_rng(seed)
    Nc = 60 + int(rng.integers(0, 30))
    Np = 60 * Nc
    p = synth.make_circle(Nc, Np, 8, spherical=spherical, focal_fixed=focal_fixed, check_in_frame=False, seed=seed, xy_range=0.25)
    nb = Np // block + 1
    kb = rng.integers(3, 9, size=nb)
    short = rng.random(nb) < 0.15                                      # some blocks change k every 16 points: no run of 32
    k_pt = kb[np.arange(Np) // block]
    fine = rng.integers(3, 9, size=Np // 16 + 1)[np.arange(Np) // 16]
    k_pt = np.where(short[np.arange(Np) // block], fine, k_pt)
    j = np.arange(len(p.obs_cam)) % 8                                  # make_circle is point-major with 8 observations per point
    keep = j < k_pt[p.obs_pt]
    pt_fixed = p.pt_fixed.copy(); pt_fixed[rng.choice(Np, size=Np // 60, replace=False)] = 1
    return dataclasses.replace(p, obs_xy=p.obs_xy[keep], obs_cam=p.obs_cam[keep], obs_pt=p.obs_pt[keep], pt_fixed=pt_fixed)


@pytest.mark.parametrize("seed,spherical,focal_fixed,backsub", [(1, False, True, "1"), (2, True, False, "1"), (3, False, False, "0"), (4, True, True, "1"), (5, False, True, None)])
def test_runs_of_every_tile_class_in_one_problem(gpu_ctx, oracle, monkeypatch, seed, spherical, focal_fixed, backsub):
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    monkeypatch.setenv("SSFM_GRAM_KMIN", "3")
    if backsub is not None:
        monkeypatch.setenv("SSFM_GRAM_BACKSUB", backsub)
    p = multi_k_problem(seed, spherical, focal_fixed)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] and abs(s["iterations"] - os_["iterations"]) <= 1 and s["pcg_iterations_total"] == 0
    assert abs(s["final_cost"] - os_["final_cost"]) <= 1e-7 * os_["final_cost"]
    assert rel_err(cams, ocams) <= 1e-5 and abs(f - of) <= 1e-5 * of
    monkeypatch.setenv("SSFM_GRAM", "0")
    c0, p0, f0, s0 = ba.optimize(gpu_ctx, p)
    assert s0["iterations"] == s["iterations"] and rel_err(cams, c0) <= 1e-7 and rel_err(pts, p0) <= 1e-7 and abs(f - f0) <= 1e-9 * f0
    monkeypatch.delenv("SSFM_GRAM")
    # the problem really covers several tile classes: ONE launch serves them (round 5, k_schur_gram_any), SSFM_GRAM_ANY=0 brings the launch per class back -- same answer
    adj = ba.BundleAdjuster(gpu_ctx, p); adj.set_profiling(True); st = adj.run(); kt = adj.kernel_times(); adj.close()
    assert kt["k_schur_gram"]["launches"] == st["num_linearizations"]
    assert kt["k_schur_pairs2"]["launches"] > 0
    monkeypatch.setenv("SSFM_GRAM_ANY", "0")
    adj = ba.BundleAdjuster(gpu_ctx, p); adj.set_profiling(True); st1 = adj.run(); kt1 = adj.kernel_times(); c1, p1, f1 = adj.download(); adj.close()
    assert kt1["k_schur_gram"]["launches"] >= 2 * st1["num_linearizations"]
    assert st1["iterations"] == s["iterations"] and rel_err(cams, c1) <= 1e-8 and rel_err(pts, p1) <= 1e-8


@pytest.mark.parametrize("max_len,spherical,focal_fixed", [(8, False, True), (8, True, False), (7, False, False), (5, True, True)])
def test_mixed_track_lengths_in_one_launch(gpu_ctx, oracle, monkeypatch, max_len, spherical, focal_fixed):
    """Round 5 (ba_kernels.h: k_schur_gram_any): tracks of 3 ... max_len cameras give signature groups of several tile classes; one launch serves them all (every wave
    picks the instantiation of its task's class).  Against the oracle, against one launch per class (SSFM_GRAM_ANY=0, rounds 3-4) and against the pair lists."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = synth.make_ragged_circle(120, 330000, 3, max_len, spherical=spherical, focal_fixed=focal_fixed)
    info = ba.plan(p)[0]
    assert info["num_observations_grouped"] >= 0.9 * info["num_observations_used"]                   # every track length from 3 on sits in groups
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
    assert abs(s["final_cost"] - os_["final_cost"]) <= 1e-9 * os_["final_cost"]
    assert rel_err(cams, ocams) <= 1e-6 and abs(f - of) <= 1e-6 * of
    assert (np.linalg.norm(pts - opts, axis=1) / np.linalg.norm(opts, axis=1)).max() <= 1e-5
    for var, val in (("SSFM_GRAM_ANY", "0"), ("SSFM_GRAM", "0")):
        monkeypatch.setenv(var, val)
        c0, p0, f0, s0 = ba.optimize(gpu_ctx, p)
        monkeypatch.delenv(var)
        assert s0["iterations"] == s["iterations"] and rel_err(cams, c0) <= 1e-8 and rel_err(pts, p0) <= 1e-8 and abs(f - f0) <= 1e-10 * f0
