"""Substructured factorisation of the reduced camera system (csrc/band_sub.h): long components cut into segments by
separators of one band width, segments factored by parallel workgroups, separators by a block-tridiagonal chain.
It is an exact factorisation in another elimination order, so the LM run must match the oracle exactly like the
single-workgroup factorisation does."""
import numpy as np
import pytest

from spherical_sfm_amd import synth

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def point_rel_err(a, b):
    return (np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-300)).max()


@pytest.mark.parametrize("spherical,focal_fixed,P,Nc,K", [(False, True, 3, 300, 6), (True, False, 2, 240, 6), (False, False, 4, 500, 6),
                                                            (False, True, 2, 120, 6)])
def test_forced_segments_match_oracle(gpu_ctx, oracle, monkeypatch, spherical, focal_fixed, P, Nc, K):
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", str(P)); monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1"); monkeypatch.setenv("SSFM_BAND_TWIST", "0")
    monkeypatch.setenv("SSFM_RING", "0")                                   # the chain of the folded band is the subject here (rings: below)
    p = synth.make_circle(Nc, 40 * Nc, K, spherical=spherical, focal_fixed=focal_fixed, check_in_frame=False, seed=77 + P)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    assert s["band_separators"] >= 1 and s["band_segments"] > s["band_separators"]
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"]
    assert s["pcg_iterations_total"] == 0                                  # the factorisation is exact: no refinement sweeps
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5 and abs(f - of) <= 1e-5 * of
    # and it equals the uncut factorisation to rounding
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", "1")
    c1, p1, f1, s1 = ba.optimize(gpu_ctx, p)
    assert s1["band_separators"] == 0 and s1["iterations"] == s["iterations"]
    assert rel_err(cams, c1) <= 1e-8 and point_rel_err(pts, p1) <= 1e-8


def test_refinement_path_with_segments(gpu_ctx, monkeypatch):
    """An impossible PCG tolerance forces refinement sweeps; with a substructured factor each sweep rebuilds and re-solves."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", "2"); monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1"); monkeypatch.setenv("SSFM_BAND_TWIST", "0")
    p = synth.make_circle(120, 4000, 6, spherical=False, focal_fixed=False, seed=5)
    cams, pts, f, s = ba.optimize(gpu_ctx, p, pcg_tolerance=1e-30, pcg_max_iterations=2)
    assert s["band_separators"] >= 1 and s["pcg_iterations_total"] >= s["num_linearizations"]
    assert np.isfinite(cams).all() and np.isfinite(pts).all()


def test_long_component_is_cut_by_default(gpu_ctx, oracle, monkeypatch):
    """1000 cameras, K = 6 -> stride 13, coprime with 1000 -> ONE ring of 1000 cameras: cut without being asked.
    Config 2 (four rings of 75) stays on one workgroup per ring."""
    from spherical_sfm_amd import ba
    info2, _, _, _ = ba.plan(synth.make_circle(300, 3000, 6, spherical=False))
    assert info2["band_separators"] == 4 and info2["band_segments"] == 8          # twisted, not cut into chains
    p = synth.make_circle(1000, 40000, 6, spherical=False, focal_fixed=True, seed=3)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    # round 5: by default the ring is laid out in its own circular order (half the band width, a CYCLE of separators solved by cyclic reduction, band_ring.h);
    # SSFM_RING=0: the Cuthill-McKee fold cut into a chain of segments (rounds 1-4)
    for ring in ("1", "0"):
        monkeypatch.setenv("SSFM_RING", ring); monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
        info, _, _, _ = ba.plan(p)
        assert info["band_separators"] >= 3 and info["band_segments"] == info["band_separators"] + (0 if ring == "1" else 1)
        assert info["band_half_width"] == (5 if ring == "1" else 10)
        cams, pts, f, s = ba.optimize(gpu_ctx, p)
        assert s["band_separators"] == info["band_separators"] and s["pcg_iterations_total"] == 0
        assert s["iterations"] == os_["iterations"] and rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5


@pytest.mark.parametrize("spherical,focal_fixed,Nc,K", [(False, True, 300, 6), (True, False, 240, 6), (False, False, 60, 6), (False, True, 210, 8)])
def test_twisted_elimination_matches_oracle_and_plain(gpu_ctx, oracle, monkeypatch, spherical, focal_fixed, Nc, K):
    """Default plan for medium components: elimination from both ends towards a separator in the middle (band_twist_plan).  Same
    LM run as the oracle, no refinement sweeps, and equal to the one-sided factorisation to rounding."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1"); monkeypatch.delenv("SSFM_BAND_TWIST", raising=False); monkeypatch.delenv("SSFM_BAND_SEGMENTS", raising=False)
    p = synth.make_circle(Nc, 40 * Nc, K, spherical=spherical, focal_fixed=focal_fixed, check_in_frame=False, seed=5 + Nc)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    assert s["band_separators"] >= 1 and s["band_segments"] == 2 * s["band_separators"]
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5 and abs(f - of) <= 1e-5 * of
    monkeypatch.setenv("SSFM_BAND_TWIST", "0")
    c1, p1, f1, s1 = ba.optimize(gpu_ctx, p)
    assert s1["band_separators"] == 0 and s1["iterations"] == s["iterations"]
    assert rel_err(cams, c1) <= 1e-8 and point_rel_err(pts, p1) <= 1e-8


def test_twisted_refinement_path(gpu_ctx, monkeypatch):
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1"); monkeypatch.delenv("SSFM_BAND_TWIST", raising=False)
    p = synth.make_circle(120, 4000, 6, spherical=False, focal_fixed=False, seed=5)
    cams, pts, f, s = ba.optimize(gpu_ctx, p, pcg_tolerance=1e-30, pcg_max_iterations=2)
    assert s["band_separators"] >= 1 and s["pcg_iterations_total"] >= s["num_linearizations"]
    assert np.isfinite(cams).all() and np.isfinite(pts).all()


@pytest.mark.parametrize("Nc,twist,merge", [(61, "1", "1"), (61, "0", "1"), (75, "1", "1"), (90, "1", "0"), (37, "1", "1")])
def test_three_dof_pairs_merged_match_oracle(gpu_ctx, oracle, monkeypatch, Nc, twist, merge):
    """Spherical BA (3-dof camera blocks): pairs of cameras merged into 6x6 block rows of the band (band_plan), odd component
    sizes (one empty slot), with and without the twisted layout, against the oracle and against the unmerged factorisation."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1"); monkeypatch.setenv("SSFM_BAND_TWIST", twist); monkeypatch.setenv("SSFM_BAND_MERGE", merge)
    p = synth.make_circle(Nc, 30 * Nc, 6, spherical=True, focal_fixed=False, check_in_frame=False, seed=100 + Nc)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["camera_dof"] == 3 and s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5 and abs(f - of) <= 1e-5 * of
    monkeypatch.setenv("SSFM_BAND_TWIST", "0"); monkeypatch.setenv("SSFM_BAND_MERGE", "0")
    c1, p1, f1, s1 = ba.optimize(gpu_ctx, p)
    if merge == "1":
        assert s["band_half_width"] == (s1["band_half_width"] + 1) // 2
    assert s1["iterations"] == s["iterations"] and rel_err(cams, c1) <= 1e-8 and point_rel_err(pts, p1) <= 1e-8


def test_long_three_dof_component_merged_and_cut(gpu_ctx, oracle, monkeypatch):
    """Spherical BA on ONE ring of 2000 cameras: 1000 merged pairs, long enough to be cut into a chain of segments (6x6 blocks)."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = synth.make_circle(2000, 24000, 6, spherical=True, focal_fixed=True, seed=8)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    for ring in ("1", "0"):                                        # ring-native layout of the merged pairs (default) / the folded chain of rounds 1-4
        monkeypatch.setenv("SSFM_RING", ring)
        cams, pts, f, s = ba.optimize(gpu_ctx, p)
        assert s["camera_dof"] == 3 and s["band_separators"] >= 2 and s["band_segments"] == s["band_separators"] + (0 if ring == "1" else 1)
        assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
        assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5


@pytest.mark.parametrize("spherical", [False, True])
def test_shared_focal_free_above_1024_cameras(gpu_ctx, oracle, monkeypatch, spherical):
    """More than 1024 cameras with the shared focal free: the two dot products of the focal step come from k_arrow_phi (contiguous
    parts summed in a fixed order) instead of being recomputed by every workgroup of k_arrow_update."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = synth.make_circle(1100, 33000, 6, spherical=spherical, focal_fixed=False, seed=12)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5 and abs(f - of) <= 1e-5 * of


@pytest.mark.parametrize("cuts", [2, 3, 4, 5, 7, 8, 9, 13, 16, 33])
def test_ring_layout_with_forced_cuts_matches_oracle(gpu_ctx, oracle, monkeypatch, cuts):
    """Round 5, band_ring.h: every shape of the cyclic reduction -- two separators (both couplings of the pair add up), odd cycles (one pair keeps its coupling across a
    step), all-tail (<= 4 separators: one launch), one and two parallel steps in front of the tail -- on ONE ring of 1000 six-dof cameras with the shared focal free
    (the focal border rides along as the second right-hand side): same LM iterations as the oracle, cameras / points / focal <= 1e-5 (observed ~1e-10)."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1"); monkeypatch.setenv("SSFM_RING_CUTS", str(cuts))
    p = synth.make_circle(1000, 30000, 6, spherical=False, focal_fixed=False, seed=21)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    assert (s["band_segments"], s["band_separators"], s["band_half_width"]) == (cuts, cuts, 5) and s["pcg_iterations_total"] == 0
    global _ring_oracle
    try: ref = _ring_oracle
    except NameError: ref = _ring_oracle = oracle.ba_solve(p)
    ocams, opts, of, os_ = ref
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"]
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5 and abs(f - of) <= 1e-5 * of


@pytest.mark.parametrize("spherical", [False, True])
def test_two_rings_and_a_short_component(gpu_ctx, oracle, monkeypatch, spherical):
    """Two rings of 2000 cameras (SURVEY 8d's stride rule at 4000 cameras, K = 8) next to nothing else, and the same with merged 3-dof pairs: ring layouts side by side."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = synth.make_circle(4000, 60000, 8, spherical=spherical, focal_fixed=True, seed=4)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    assert s["band_half_width"] == (4 if spherical else 7) and s["band_segments"] == s["band_separators"] >= 8
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5


def _concat_problems(a, b):
    """two independent scenes as one problem: cameras and points of b renumbered behind a's (two connected components of the camera graph)"""
    import dataclasses
    na, pa = len(a.cameras), len(a.points)
    return dataclasses.replace(a, cameras=np.concatenate([a.cameras, b.cameras]), points=np.concatenate([a.points, b.points]),
                               obs_xy=np.concatenate([a.obs_xy, b.obs_xy]), obs_cam=np.concatenate([a.obs_cam, b.obs_cam + na]).astype(np.int32),
                               obs_pt=np.concatenate([a.obs_pt, b.obs_pt + pa]).astype(np.int32), rot_fixed=np.concatenate([a.rot_fixed, b.rot_fixed]),
                               trans_fixed=np.concatenate([a.trans_fixed, b.trans_fixed]), pt_fixed=np.concatenate([a.pt_fixed, b.pt_fixed]),
                               gt_cameras=np.concatenate([a.gt_cameras, b.gt_cameras]), gt_points=np.concatenate([a.gt_points, b.gt_points]))


def test_ring_next_to_short_components(gpu_ctx, oracle, monkeypatch):
    """A long ring (1000 cameras, reach 5: ring layout) in one problem with config 2's four short rings (75 cameras each, folded half-width 10: twisted).  The band has ONE
    half-width -- 10, the widest component's -- so the long ring is laid out with separators of 10 rows although 5 would do; a third scene whose fold is wider than the
    cyclic-reduction kernel's blocks allow (K = 10: half-width 18 x 6 = 108 > 78) makes the planner withdraw the ring layout and cut the fold into a chain as before."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    long_ring = synth.make_circle(1000, 20000, 6, spherical=False, focal_fixed=True, seed=3)
    short = synth.make_circle(300, 6000, 6, spherical=False, focal_fixed=True, seed=4)
    p = _concat_problems(long_ring, short)
    info = ba.plan(p)[0]
    assert info["band_half_width"] == 10
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    assert s["band_segments"] - 8 == s["band_separators"] - 4 >= 4            # (8 twisted halves + 4 separators of the short rings; the rest: arcs = separators of the long ring)
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5
    wide = synth.make_circle(150, 6000, 10, spherical=False, focal_fixed=True, seed=5, check_in_frame=False, xy_range=0.2)
    p2 = _concat_problems(long_ring, wide)
    info2 = ba.plan(p2)[0]
    assert info2["band_half_width"] * 6 > 78 and info2["band_segments"] != info2["band_separators"]     # no ring layout at that width
    cams, pts, f, s = ba.optimize(gpu_ctx, p2)
    ocams, opts, of, os_ = oracle.ba_solve(p2)
    assert s["iterations"] == os_["iterations"] and rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5


def test_odd_merged_ring_and_refinement_path(gpu_ctx, oracle, monkeypatch):
    """Spherical BA on a ring of 2003 cameras: 1002 merged pairs, the last one with an empty slot that lies in the ring's LAST separator (the one with the copy slot in
    front of the first arc); then the same with an impossible PCG tolerance, so that every LM iteration rebuilds the band and runs the ring solve again on the residual."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = synth.make_circle(2003, 24000, 6, spherical=True, focal_fixed=False, seed=8)
    cams, pts, f, s = ba.optimize(gpu_ctx, p)
    assert s["camera_dof"] == 3 and s["band_segments"] == s["band_separators"] >= 4
    ocams, opts, of, os_ = oracle.ba_solve(p)
    assert s["termination"] == os_["termination"] == 0 and s["iterations"] == os_["iterations"] and s["pcg_iterations_total"] == 0
    assert rel_err(cams, ocams) <= 1e-5 and point_rel_err(pts, opts) <= 1e-5 and abs(f - of) <= 1e-5 * of
    c2, p2, f2, s2 = ba.optimize(gpu_ctx, p, pcg_tolerance=1e-30, pcg_max_iterations=2)
    assert s2["pcg_iterations_total"] >= s2["num_linearizations"] and np.isfinite(c2).all() and np.isfinite(p2).all()
    assert rel_err(c2, ocams) <= 1e-5 and point_rel_err(p2, opts) <= 1e-5
