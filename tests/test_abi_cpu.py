"""CPU tests of the C-ABI library itself: it loads, exports every symbol include/ssfm.h declares, refuses to run
without a GPU (no CPU fallback), and its host-side planning (flatten rules, sharding, elimination order) is right.
No compute kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from spherical_sfm_amd import _lib, ba, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "ssfm.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ssfm_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = header_functions()
    assert declared, "no functions parsed from include/ssfm.h"
    for name in declared:
        assert hasattr(L, name), f"libssfm_hip.so does not export {name}"
    assert sorted(_lib.DECLARED_SYMBOLS) == declared
    assert L.ssfm_version() >= 100


def test_default_options_are_the_reference_values():
    o = ba.default_options()
    assert o.max_num_iterations == 2000 and o.max_num_consecutive_invalid_steps == 100      # src/sfm.cpp:205-206
    assert o.loss_type == 1 and o.loss_scale == 1.0                                          # CauchyLoss(1.0), src/sfm.cpp:196
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)   # Ceres 2.2.0 defaults
    assert (o.initial_trust_region_radius, o.min_relative_decrease, o.jacobi_scaling) == (1e4, 1e-3, 1)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.SsfmError, match="no HIP device"):
        ba.Context(0)


def test_struct_sizes_match_the_header():
    # 4+4+8 + 9 pointers*8 + 4 (+4 pad)
    assert C.sizeof(_lib.BAProblemC) == 96
    assert C.sizeof(_lib.BAOptionsC) % 8 == 0 and C.sizeof(_lib.BASummaryC) % 8 == 0


def test_plan_applies_the_reference_flatten_rules():
    p = synth.make_circle(60, 400, 6)
    p.points[11] = 0.0                                                   # |X| == 0 (src/sfm.cpp:243)
    keep = ~((p.obs_pt == 20) & (np.arange(len(p.obs_pt)) % 6 >= 2))     # 2 observations left (src/sfm.cpp:254)
    p.obs_xy, p.obs_cam, p.obs_pt = p.obs_xy[keep], p.obs_cam[keep], p.obs_pt[keep]
    info, ids, used, pos = ba.plan(p)
    assert info["num_points_used"] == 398 and info["num_observations_used"] == 6 * 398
    assert 11 not in ids and 20 not in ids and (np.diff(ids) > 0).all()
    assert not used[p.obs_pt == 11].any() and not used[p.obs_pt == 20].any() and used.sum() == 6 * 398
    assert info["camera_dof"] == 3
    p2 = synth.make_circle(60, 400, 6, spherical=False)
    assert ba.plan(p2)[0]["camera_dof"] == 6


def test_plan_accepts_unsorted_input_and_keeps_last_duplicate():
    p = synth.make_circle(60, 100, 6)
    rng = np.random.default_rng(3)
    j = int(np.where(p.obs_pt == 5)[0][0])
    p.obs_xy = np.vstack([p.obs_xy, p.obs_xy[j] + 1.0]); p.obs_cam = np.append(p.obs_cam, p.obs_cam[j]); p.obs_pt = np.append(p.obs_pt, 5)
    n = len(p.obs_pt)
    perm = np.concatenate([rng.permutation(n - 1), [n - 1]])
    p.obs_xy, p.obs_cam, p.obs_pt = p.obs_xy[perm], p.obs_cam[perm].astype(np.int32), p.obs_pt[perm].astype(np.int32)
    info, ids, used, pos = ba.plan(p)
    assert info["num_observations_used"] == 600 and used.sum() == 600
    assert used[-1] == 1 and used[np.where(perm == j)[0][0]] == 0       # the later duplicate wins (std::map assignment)


@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_sharding_partitions_the_used_points(nranks):
    p = synth.make_circle(60, 1000, 6, spherical=False)
    p.points[::97] = 0.0
    all_ids, total_obs, sizes = [], 0, []
    for r in range(nranks):
        info, ids, used, pos = ba.plan(p, nranks, r)
        assert info["num_points_used_global"] == len(p.points) - len(p.points[::97])
        all_ids.append(ids); total_obs += info["num_observations_used"]; sizes.append(info["num_observations_used"])
        assert used.sum() == info["num_observations_used"]
        # every rank keeps whole points: all 6 observations of an owned point
        assert (np.bincount(p.obs_pt[used.astype(bool)], minlength=len(p.points))[ids] == 6).all()
    cat = np.concatenate(all_ids)
    assert len(np.unique(cat)) == len(cat) == info["num_points_used_global"] and (np.diff(cat) > 0).all()   # disjoint, contiguous ranges
    assert total_obs == info["num_observations_used_global"]
    assert max(sizes) - min(sizes) <= 6 * 2                               # balanced by observation count


def test_elimination_order_is_a_banded_permutation(monkeypatch):
    p = synth.make_circle(60, 600, 6, spherical=False)           # 6-dof camera blocks: band rows = cameras
    cams = p.obs_cam.reshape(-1, 6)
    monkeypatch.setenv("SSFM_BAND_TWIST", "0")                    # plain Cuthill-McKee order
    info, ids, used, pos = ba.plan(p)
    assert sorted(pos.tolist()) == list(range(60))
    # cameras that share a point must sit within the reported half-bandwidth of each other
    d = np.abs(pos[cams][:, :, None] - pos[cams][:, None, :]).max()
    assert d == info["band_half_width"]
    assert info["band_half_width"] <= 12          # a ring with reach 5 folds into a band of ~2*5
    assert info["reduced_blocks"] == 60 * 6 and info["max_row_blocks"] <= 11      # lower triangle: 10 neighbours / 2 + diagonal
    assert (info["band_segments"], info["band_separators"]) == (1, 0)
    # default: the ring is eliminated from both ends (csrc/ba_flatten.h: band_twist_plan): seg_0 | seg_1 reversed | separator last
    monkeypatch.delenv("SSFM_BAND_TWIST")
    info2, _, _, pos2 = ba.plan(p)
    b = info2["band_half_width"]; m0 = (60 - b) // 2; m1 = 60 - b - m0
    assert sorted(pos2.tolist()) == list(range(60)) and b == info["band_half_width"]
    assert (info2["band_segments"], info2["band_separators"]) == (2, 1) and info2["reduced_blocks"] == 60 * 6
    seg = np.where(pos2 < m0, 0, np.where(pos2 < m0 + m1, 1, 2))      # 2 = separator
    a, c = pos2[cams][:, :, None], pos2[cams][:, None, :]
    sa, sc = seg[cams][:, :, None], seg[cams][:, None, :]
    assert not ((sa == 0) & (sc == 1)).any()                          # the separator really separates the two segments
    same = (sa == sc) & (sa < 2)
    assert np.abs(a - c)[same].max() <= b                             # each segment is still a band of the same width
    assert np.array_equal(np.sort(pos2[pos < m0]), np.arange(m0))     # seg_0 = the first m0 Cuthill-McKee positions, in the same order
    assert np.array_equal(pos2[pos < m0], pos[pos < m0])
    tail = pos >= m0 + b                                              # seg_1 = the last m1 positions, reversed
    assert np.array_equal(pos2[tail], m0 + (m1 - 1 - (pos[tail] - m0 - b)))


def test_three_dof_cameras_are_merged_in_pairs(monkeypatch):
    """Spherical BA (3-dof camera blocks): two consecutive cameras of the Cuthill-McKee order share one 6x6 block row of the band
    (csrc/ba_flatten.h: band_plan), so the band is half as long and about half as wide in blocks."""
    p = synth.make_circle(61, 610, 6)                             # odd: the last pair has an empty slot
    cams = p.obs_cam.reshape(-1, 6)
    monkeypatch.setenv("SSFM_BAND_MERGE", "0"); monkeypatch.setenv("SSFM_BAND_TWIST", "0")
    plain, _, _, pos0 = ba.plan(p)
    monkeypatch.delenv("SSFM_BAND_MERGE")
    merged, _, _, pos1 = ba.plan(p)
    assert plain["camera_dof"] == merged["camera_dof"] == 3
    assert sorted(pos1.tolist()) == list(range(61)) and np.array_equal(pos0, pos1)         # same order, paired up
    assert merged["band_half_width"] == (plain["band_half_width"] + 1) // 2
    d = np.abs(pos1[cams][:, :, None] // 2 - pos1[cams][:, None, :] // 2).max()
    assert d == merged["band_half_width"]
    monkeypatch.delenv("SSFM_BAND_TWIST")
    tw, _, _, pos2 = ba.plan(p)
    assert (tw["band_segments"], tw["band_separators"]) == (2, 1) and sorted(pos2.tolist()) == list(range(61))


def test_config2_plan():
    """BASELINE configs[1] sizes; the stride-4 recipe of SURVEY 8d gives four interleaved camera rings."""
    p = synth.make_circle(300, 100000, 6, spherical=False)
    info, ids, used, pos = ba.plan(p)
    assert info["num_observations_used_global"] == 600000 and info["num_points_used_global"] == 100000
    assert info["reduced_blocks"] == 300 * 6 and info["band_half_width"] <= 12


def test_signature_groups_of_the_plan(monkeypatch):
    """Runs of >= 32 consecutive points with the same 3..8 cameras are assembled as Gram products; the plan reports how many."""
    import dataclasses
    monkeypatch.delenv("SSFM_GRAM", raising=False); monkeypatch.delenv("SSFM_GRAM_KMIN", raising=False); monkeypatch.delenv("SSFM_GRAM_PTS", raising=False)
    assert C.sizeof(_lib.BAPlanInfoC) == 72
    p = synth.make_circle(300, 100000, 6, spherical=False)                 # config 2: every point of a camera window has the same six cameras
    info = ba.plan(p)[0]
    assert info["num_points_grouped"] == 100000 and info["num_observations_grouped"] == 600000
    assert info["group_tasks"] == 900                                       # 300 runs of 333..334 points, each cut in 3 equal tasks (round 4: at most 4 x CUs tasks, one wave per SIMD)
    parts = [ba.plan(p, 4, r)[0] for r in range(4)]                        # sharded: every rank groups its own points
    assert sum(q["num_points_grouped"] for q in parts) >= 100000 - 4 * 4 * 32 and all(q["num_points_grouped"] <= q["num_points_used"] for q in parts)
    # ten cameras per point: more than a task holds; SSFM_GRAM_KMIN: a lower bound on the camera list; SSFM_GRAM=0: off
    assert ba.plan(synth.make_circle(120, 4800, 10, spherical=True, check_in_frame=False, xy_range=0.2))[0]["num_points_grouped"] == 0
    assert ba.plan(synth.make_circle(60, 4200, 3, spherical=False))[0]["num_points_grouped"] == 4200
    monkeypatch.setenv("SSFM_GRAM_KMIN", "4")
    assert ba.plan(synth.make_circle(60, 4200, 3, spherical=False))[0]["num_points_grouped"] == 0
    monkeypatch.delenv("SSFM_GRAM_KMIN")
    # a loose point every 20 points: no run reaches 32
    q = synth.make_circle(60, 4200, 6, spherical=False)
    keep = ~((q.obs_pt % 20 == 0) & (np.arange(len(q.obs_pt)) % 6 == 5))
    q2 = dataclasses.replace(q, obs_xy=q.obs_xy[keep], obs_cam=q.obs_cam[keep], obs_pt=q.obs_pt[keep])
    assert ba.plan(q2)[0]["num_points_grouped"] == 0 and ba.plan(q)[0]["num_points_grouped"] == 4200
    monkeypatch.setenv("SSFM_GRAM", "0")
    assert ba.plan(p)[0]["num_points_grouped"] == 0


def test_band_segment_plan_matches_reference_partition(monkeypatch):
    """The segment / separator tables of the substructured factorisation (csrc/band_sub.h) are host logic: ssfm_ba_plan reports
    their sizes; tests/_band_ref.py restates the partition rule."""
    import _band_ref as R
    from spherical_sfm_amd import ba, synth
    p = synth.make_circle(1000, 4000, 6, spherical=False, focal_fixed=True)          # stride 13, coprime with 1000: one ring of 1000
    monkeypatch.delenv("SSFM_BAND_SEGMENTS", raising=False)
    monkeypatch.setenv("SSFM_RING", "0")                                             # the Cuthill-McKee fold cut into a chain (rounds 1-4; rings: test_ring_layout_of_long_rings)
    info, _, _, pos = ba.plan(p)
    assert info["band_separators"] >= 3 and info["band_segments"] == info["band_separators"] + 1        # cut without being asked
    b = info["band_half_width"]
    for P in (2, 5, 64):
        monkeypatch.setenv("SSFM_BAND_SEGMENTS", str(P))
        got, _, _, _ = ba.plan(p)
        segs, seps = R.segment_table([0, 1000], b, P)
        assert (got["band_segments"], got["band_separators"]) == (len(segs), len(seps))
        assert all(hi - lo >= b + 1 for lo, hi, _, _ in segs) and sum(hi - lo for lo, hi, _, _ in segs) + b * len(seps) == 1000
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", "1")
    assert ba.plan(p)[0]["band_separators"] == 0
    monkeypatch.delenv("SSFM_BAND_SEGMENTS")
    monkeypatch.setenv("SSFM_BAND_TWIST", "0")
    small, _, _, _ = ba.plan(synth.make_circle(300, 3000, 6, spherical=False))       # config 2 shape: four rings of 75 are not cut into chains ...
    assert (small["band_segments"], small["band_separators"]) == (4, 0)
    monkeypatch.delenv("SSFM_BAND_TWIST")
    small, _, _, _ = ba.plan(synth.make_circle(300, 3000, 6, spherical=False))       # ... but eliminated from both ends
    assert (small["band_segments"], small["band_separators"]) == (8, 4)


def test_ring_components_fold_to_twice_their_reach(monkeypatch):
    """Camera ordering (csrc/ba_flatten.h: cuthill_mckee): a ring whose cameras share points with their +-r neighbours is a periodic band of
    half-width r; folded into a plain band it cannot be narrower than 2 r, and the planner reaches that (the root's children go closest first)."""
    for nc, k, reach in ((300, 6, 5), (75, 6, 5), (60, 6, 5), (500, 6, 5), (4000, 8, 7)):
        p = synth.make_circle(nc, 2000 if nc < 1000 else 16000, k, spherical=False, focal_fixed=True)
        if nc >= 500:  monkeypatch.setenv("SSFM_RING", "0")                           # longer rings are no longer folded by default (next test)
        info = ba.plan(p)[0]
        assert info["band_half_width"] == 2 * reach, (nc, k, info["band_half_width"])
    monkeypatch.delenv("SSFM_RING", raising=False)
    p = synth.make_circle(300, 2000, 6, spherical=True, focal_fixed=True)      # 3-dof cameras: pairs merged into 6x6 band blocks
    assert ba.plan(p)[0]["band_half_width"] == 5


def test_ring_layout_of_long_rings(monkeypatch):
    """Round 5 (csrc/ba_flatten.h: band_plan, RingComp): a long camera ring is laid out in its own circular order -- a periodic band of half-width = its reach, HALF of
    what the Cuthill-McKee fold needs -- as  copy of S_{m-1} | A_0 | S_0 | ... | A_{m-1} | S_{m-1}:  m arcs (segments) and m separators of `reach` block rows.
    The circular order is found from the graph (ids strided as in SURVEY 8d's circle: a walk round the ring) or is the id order (video-like tracks with a loop closure)."""
    monkeypatch.delenv("SSFM_RING", raising=False); monkeypatch.delenv("SSFM_RING_CUTS", raising=False)
    # (cameras, points, K, spherical) -> reach in block rows
    for nc, npts, k, sph, reach, rings in ((1000, 4000, 6, False, 5, 1), (4000, 16000, 8, False, 7, 2), (4000, 16000, 8, True, 4, 2)):
        p = synth.make_circle(nc, npts, k, spherical=sph, focal_fixed=True)
        info, _, _, pos = ba.plan(p)
        assert info["band_half_width"] == reach and sorted(pos.tolist()) == list(range(nc))
        assert info["band_segments"] == info["band_separators"] and info["band_segments"] >= 2 * rings          # a cycle: as many separators as arcs
        # the elimination order is the circular order: coupled cameras are within `reach` block rows of each other, modulo the ring
        cams = p.obs_cam.reshape(-1, k); w = 2 if sph else 1
        ring_of = np.zeros(nc, int); rows = nc // rings // w
        order = np.argsort(pos); ring_of[order] = np.arange(nc) // (nc // rings)                              # rings are consecutive position ranges
        a = pos[cams][:, :, None] // w; c = pos[cams][:, None, :] // w
        d = np.abs(a - c); d = np.minimum(d, rows - d)
        assert d.max() == reach and (ring_of[cams] == ring_of[cams][:, :1]).all()
    # forced number of cuts; video-like ragged tracks (ids in circular order already, a band too wide for the LDS window when folded)
    monkeypatch.setenv("SSFM_RING_CUTS", "4")
    info = ba.plan(synth.make_circle(1000, 4000, 6, spherical=False, focal_fixed=True))[0]
    assert (info["band_segments"], info["band_separators"]) == (4, 4)
    monkeypatch.delenv("SSFM_RING_CUTS")
    rag = synth.make_ragged_circle(300, 60000, 3, 14)
    info = ba.plan(rag)[0]
    assert info["band_half_width"] == 13 and info["band_segments"] == info["band_separators"] >= 2
    monkeypatch.setenv("SSFM_RING", "0")
    assert ba.plan(rag)[0]["band_half_width"] == 26
    monkeypatch.delenv("SSFM_RING")
    # short rings stay folded and twisted when the cost model says so (config 2: four rings of 75) ...
    info = ba.plan(synth.make_circle(300, 3000, 6, spherical=False))[0]
    assert info["band_half_width"] == 10 and (info["band_segments"], info["band_separators"]) == (8, 4)
    # ... and go to the ring layout when it does not: one ring of 300 cameras with reach 7 (tracks of 3 ... 8 cameras), one of 500 with reach 5
    info = ba.plan(synth.make_ragged_circle(300, 60000, 3, 8))[0]
    assert info["band_half_width"] == 7 and (info["band_segments"], info["band_separators"]) == (8, 8)
    info = ba.plan(synth.make_circle(500, 2000, 6, spherical=False, focal_fixed=True))[0]
    assert info["band_half_width"] == 5 and (info["band_segments"], info["band_separators"]) == (16, 16)


def test_planner_pool_survives_fork():
    """The planner's helper threads do not exist in a fork()ed child; the child must start its own pool instead of waiting for them."""
    import os
    import signal
    p = synth.make_circle(300, 50000, 6, spherical=False, focal_fixed=True)       # large enough for the threaded sections
    want = ba.plan(p)[0]["band_half_width"]
    pid = os.fork()
    if pid == 0:
        signal.alarm(30)                                                         # a deadlock ends the child with SIGALRM
        ok = ba.plan(p)[0]["band_half_width"] == want
        os._exit(0 if ok else 1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status


def test_signature_sort_groups_scattered_tracks(monkeypatch):
    """Round 4: points whose camera lists are equal but whose ids are apart (build_sfm issues ids in match order: tracks of different length that start in the same
    frame are interleaved, examples/spherical_sfm_tools.cpp:862-955) are made adjacent by the planner's signature sort, so they can share wave tasks of k_schur_gram;
    the cost model then decides whether the groups are used."""
    from spherical_sfm_amd import ba, synth
    prob = synth.make_ragged_circle(100, 200000, 3, 8, seed=9)
    monkeypatch.setenv("SSFM_GRAM_MODEL", "0")
    monkeypatch.setenv("SSFM_GRAM_SORT", "0")
    d0, ids0, used0, _ = ba.plan(prob)
    monkeypatch.delenv("SSFM_GRAM_SORT")
    d1, ids1, used1, _ = ba.plan(prob)
    assert d0["num_points_grouped"] == 0 and (np.diff(ids0) > 0).all()                   # interleaved lengths: no run of 32 equal lists in the caller's order
    assert d1["num_points_grouped"] >= 0.95 * d1["num_points_used"]                      # 100 start frames x 6 lengths = 600 signatures of ~60 points
    assert np.array_equal(np.sort(ids1), ids0) and np.array_equal(used0, used1)          # a permutation of the same points, the same observations
    assert d1["band_half_width"] == d0["band_half_width"] and d1["reduced_blocks"] == d0["reduced_blocks"]
    monkeypatch.delenv("SSFM_GRAM_MODEL")
    d2, ids2, _, _ = ba.plan(prob)
    assert d2["num_points_grouped"] in (0, d1["num_points_grouped"])                     # the model keeps all groups or none
    # signatures with fewer members than a wave task wants: the caller's order stays
    small = synth.make_ragged_circle(150, 12000, 3, 14, seed=9)
    d3, ids3, _, _ = ba.plan(small)
    assert (np.diff(ids3) > 0).all() and d3["num_points_grouped"] == 0
