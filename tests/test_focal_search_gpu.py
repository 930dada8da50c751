"""Batched focal-length search (ssfm_focal_search) against the reference's loop_constraint_cost_fn restated with the oracle's
pieces (examples/spherical_sfm_tools.cpp:794-813,1118-1157,1418-1496; get_cost src/uncalibrated_pose_graph.cpp:116-145)."""
import numpy as np
import pytest

from spherical_sfm_amd import rotavg, synth
from _uncalib_graph import make_uncalibrated_loop

pytestmark = pytest.mark.gpu


def _oracle_cost(O, n, i0, i1, R_rel, focal, focal_guess, inward=False):
    T = np.diag([focal / focal_guess, focal / focal_guess, 1.0])
    Rn = np.zeros_like(R_rel)
    for k in range(len(i0)):                                                     # transform_image_matches
        E = O.make_spherical_essential_matrix(R_rel[k], inward)
        r, _ = O.decompose_spherical_essential_matrix(T @ E @ T, inward)
        Rn[k] = synth.so3exp(np.asarray(r)[None])[0]
    rot = np.tile(np.eye(3), (n, 1, 1)); R = np.eye(3)                          # initialize_rotations_sequential
    for idx in range(1, n):
        for k in range(len(i0)):
            if i0[k] == idx - 1 and i1[k] == idx:
                R = Rn[k] @ R; rot[idx] = R
                break
    return O.get_cost(rot, i0, i1, Rn), rot


def test_costs_argmin_and_rotations_match_the_oracle(gpu_ctx, oracle):
    n = 40
    i0, i1, R_rel, R_gt = make_uncalibrated_loop(oracle, n, 3, focal_true=1000.0, focal_guess=1300.0)
    focals = np.random.default_rng(3).uniform(1300.0 / 4, 1300.0 * 2, 48)       # min/max as run_spherical_sfm_uncalib.cpp:141-142
    costs, best, rot = rotavg.focal_search(gpu_ctx, n, i0, i1, R_rel, 1300.0, focals)
    ref = []; rots = []
    for f in focals:
        c, r = _oracle_cost(oracle, n, i0, i1, R_rel, f, 1300.0)
        ref.append(c); rots.append(r)
    ref = np.array(ref)
    assert np.abs(costs - ref).max() <= 1e-9 * ref.max()
    assert best == int(np.argmin(ref))
    assert np.abs(rot - rots[best]).max() < 1e-10
    # the loop closes best near the true focal
    assert abs(focals[best] - 1000.0) < 0.08 * 1000.0


def test_search_then_optimisation_recovers_the_focal(gpu_ctx, oracle):
    """find_best_focal_length_random end to end: trials -> best -> optimize_rotations_and_focal_length (run_optimization)."""
    n = 60
    i0, i1, R_rel, R_gt = make_uncalibrated_loop(oracle, n, 3, focal_true=1000.0, focal_guess=1250.0, noise_deg=0.02)
    focals = np.random.default_rng(11).uniform(1250.0 / 4, 1250.0 * 2, 256)
    costs, best, rot = rotavg.focal_search(gpu_ctx, n, i0, i1, R_rel, 1250.0, focals)
    assert abs(focals[best] - 1000.0) < 60.0
    # run_optimization (tools.cpp:1160-1188): matches re-derived at the best focal, then the joint solve with box bounds
    costs, best, rot, rel_best = rotavg.focal_search(gpu_ctx, n, i0, i1, R_rel, 1250.0, focals, return_matches=True)
    f0 = float(focals[best])
    rot2, f_opt, cost, summ = rotavg.optimize_rotations_and_focal_length(gpu_ctx, rot, i0, i1, rel_best, f0, 1250.0 / 4, 1250.0 * 2)
    assert summ["termination"] in (0, 1) and cost <= costs[best] * (1 + 1e-9)
    assert abs(f_opt - 1000.0) <= abs(f0 - 1000.0) + 5.0 and abs(f_opt - 1000.0) < 40.0


def test_missing_chain_edges_and_single_trial(gpu_ctx, oracle):
    n = 12
    i0, i1, R_rel, _ = make_uncalibrated_loop(oracle, n, 2, focal_true=900.0, focal_guess=1000.0)
    keep = ~((i0 == 4) & (i1 == 5))                                               # camera 5 has no (4,5) match: stays identity, chain continues
    i0, i1, R_rel = i0[keep], i1[keep], R_rel[keep]
    costs, best, rot = rotavg.focal_search(gpu_ctx, n, i0, i1, R_rel, 1000.0, [950.0])
    c, r = _oracle_cost(oracle, n, i0, i1, R_rel, 950.0, 1000.0)
    assert best == 0 and abs(costs[0] - c) <= 1e-9 * c and np.abs(rot - r).max() < 1e-10
    assert np.array_equal(rot[5], np.eye(3)) and not np.array_equal(rot[6], np.eye(3))
