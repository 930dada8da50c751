"""The supernodal ring / chain solver of the reduced camera system alone (ssfm_snode_solve_probe, csrc/snode.h): random symmetric positive definite
block-sparse systems with the structure tracks give the reduced matrix -- every window of r + 1 consecutive cameras of a ring (or chain) is fully
coupled -- against numpy's dense solve.  Rings whose length is and is not a multiple of the supernode, several interleaved rings (the camera ids of
BASELINE config 2: ring = id mod 4), chains down to one camera, both camera block sizes, one and two right-hand sides, blocks stored in either
orientation; structures the plan must refuse."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def window_system(comps, dc, seed, shuffle_ids=False):
    """comps: list of (n cameras, reach r, ring?).  Cameras of component k get the ids k, k + K, k + 2K, ... (interleaved) or a random relabelling.
    Returns (row_ptr, col_idx, blocks, dense A, rhs2)."""
    rng = np.random.default_rng(seed)
    K = len(comps); total = sum(c[0] for c in comps)
    ids = []
    if shuffle_ids:
        perm = rng.permutation(total); at = 0
        for n, r, ring in comps:
            ids.append(perm[at:at + n]); at += n
    else:                                                        # interleave: component k owns ids k, k + K, ... while it lasts, the rest follow
        pool = list(range(total)); nxt = [[] for _ in comps]; left = [c[0] for c in comps]; k = 0
        for i in pool:
            while left[k % K] == 0:
                k += 1
            nxt[k % K].append(i); left[k % K] -= 1; k += 1
        ids = [np.array(v) for v in nxt]
    N = total * dc
    A = np.zeros((N, N))
    coupled = set()
    for (n, r, ring), cid in zip(comps, ids):
        starts = range(n) if ring else range(max(1, n - r))
        for s0 in starts:
            cams = [cid[(s0 + t) % n] for t in range(min(r + 1, n))] if ring else [cid[t] for t in range(s0, min(s0 + r + 1, n))]
            cams = list(dict.fromkeys(cams))
            dof = np.concatenate([np.arange(c * dc, (c + 1) * dc) for c in cams])
            X = rng.normal(size=(len(dof), 4))
            A[np.ix_(dof, dof)] += X @ X.T
            for a in cams:
                for b in cams:
                    if a != b:
                        coupled.add((min(a, b), max(a, b)))
    A += np.diag(rng.uniform(0.5, 1.5, N))
    rows = [[] for _ in range(total)]
    for c in range(total):
        rows[c].append(c)
    for a, b in sorted(coupled):
        if rng.random() < 0.5:
            rows[a].append(b)
        else:
            rows[b].append(a)
    row_ptr = [0]; col_idx = []; blocks = []
    for c in range(total):
        for c2 in sorted(rows[c]):
            col_idx.append(c2); blocks.append(A[c * dc:(c + 1) * dc, c2 * dc:(c2 + 1) * dc].copy())
        row_ptr.append(len(col_idx))
    rhs2 = rng.normal(size=(2, N))
    return np.array(row_ptr, np.int32), np.array(col_idx, np.int32), np.array(blocks), A, rhs2


CASES = [
    (6, [(75, 5, True)] * 4, False),                 # BASELINE config 2: four rings of 75 cameras, tracks of six frames, ids interleaved
    (6, [(72, 5, True), (79, 5, True)], False),      # T of 7 and of 9 cameras (42 / 54 rows: the second panel of T)
    (6, [(60, 5, True)], False),                     # BASELINE config 1
    (6, [(71, 5, True), (72, 5, True)], True),       # ids shuffled: the greedy walk has to find the circular order
    (6, [(40, 3, True), (23, 5, False), (1, 0, False), (5, 4, False), (11, 5, False), (17, 2, False)], False),
    (3, [(120, 10, True), (95, 7, True)], False),
    (3, [(44, 10, True), (9, 3, False), (31, 10, False)], True),
    (6, [(20, 5, True)], False),                     # the smallest ring the plan takes (three supernodes + T)
    (6, [(27, 1, True), (6, 5, False)], False),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("nr", [2, 1])
def test_snode_solver_matches_dense_solve(gpu_ctx, case, nr):
    from spherical_sfm_amd import ba
    dc, comps, shuffle = CASES[case]
    rp, ci, blk, A, rhs2 = window_system(comps, dc, seed=100 + case, shuffle_ids=shuffle)
    Y, info = ba.snode_solve_probe(gpu_ctx, dc, rp, ci, blk, rhs2, nr=nr)
    assert info["applies"] and info["failed"] == 0, info
    xr = np.linalg.solve(A, rhs2.T).T
    assert np.abs(Y[:nr] - xr[:nr]).max() <= 1e-12 * np.abs(xr).max(), (np.abs(Y[:nr] - xr[:nr]).max(), info)
    if nr == 1:
        assert not Y[1].any()                                   # the second column is not touched
    Y2, _ = ba.snode_solve_probe(gpu_ctx, dc, rp, ci, blk, rhs2, nr=nr)
    assert np.array_equal(Y, Y2)                               # no atomics, fixed order: the same bits every time


@pytest.mark.parametrize("dc,comps", [(6, [(75, 7, True)]), (6, [(19, 5, True)]), (3, [(80, 11, True)]), (6, [(400, 5, True)])])
def test_snode_plan_refuses_what_it_cannot_take(gpu_ctx, dc, comps):
    """reach beyond 30 / dc cameras, rings under four supernodes (dense for this purpose), halves longer than the step table"""
    from spherical_sfm_amd import ba
    rp, ci, blk, A, rhs2 = window_system(comps, dc, seed=7)
    Y, info = ba.snode_solve_probe(gpu_ctx, dc, rp, ci, blk, rhs2)
    if comps[0][0] == 19:                                       # 19 cameras with reach 5: every order is a chain of reach > 5 -> refused; (reach 5 rings need >= 20)
        assert not info["applies"]
    else:
        assert not info["applies"] and not Y.any()


def test_snode_failure_flag(gpu_ctx):
    from spherical_sfm_amd import ba
    rp, ci, blk, A, rhs2 = window_system([(75, 5, True)], 6, seed=3)
    d = [k for k in range(len(ci)) if ci[k] == 40 and rp[40] <= k < rp[41]][0]
    blk[d] -= np.eye(6) * 1e4
    assert ba.snode_solve_probe(gpu_ctx, 6, rp, ci, blk, rhs2)[1]["failed"] == 1


@pytest.mark.parametrize("cams,spherical,focal_fixed", [(60, False, True), (60, True, False), (300, False, False), (300, False, True), (120, True, True)])
def test_lm_loop_with_the_supernodal_solver_matches_the_oracle(gpu_ctx, oracle, monkeypatch, cams, spherical, focal_fixed):
    """SSFM_SNODE=1: the LM loop's direct solve through k_snode_solve (opt-in, csrc/snode.h) -- same iteration count and the same answer as the oracle's
    sparse Cholesky, one and two right-hand sides (free focal), 6- and 3-dof cameras, one ring (stride 1) and four interleaved rings (BASELINE config 2's ids)."""
    from spherical_sfm_amd import ba, synth
    monkeypatch.setenv("SSFM_SNODE", "1")
    monkeypatch.setenv("SSFM_NO_PLAN_CACHE", "1")
    p = synth.make_circle(cams, cams * 40, 6, spherical=spherical, focal_fixed=focal_fixed)
    c1, p1, f1, s1 = ba.optimize(gpu_ctx, p)
    oc, op, of, os_ = oracle.ba_solve(p)
    assert s1["termination"] == os_["termination"] == 0 and s1["iterations"] == os_["iterations"]
    assert np.abs(c1 - oc).max() <= 1e-5 * np.abs(oc).max() and abs(f1 - of) <= 1e-5 * of
    assert (np.linalg.norm(p1 - op, axis=1) / np.linalg.norm(op, axis=1)).max() <= 1e-5
    monkeypatch.setenv("SSFM_SNODE", "0")
    c0, p0, f0, s0 = ba.optimize(gpu_ctx, p)                   # the band kernels on the same problem: two direct solvers, one answer
    assert s0["iterations"] == s1["iterations"] and np.abs(c0 - c1).max() <= 1e-9 * np.abs(c0).max()
