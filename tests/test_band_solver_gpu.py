"""The reduced-system solver alone (ssfm_band_solve_probe): random symmetric positive definite block bands against numpy's dense
solve, single-workgroup factorisation and the substructured one (csrc/band_sub.h), both camera block sizes, several components."""
import numpy as np
import pytest

import _band_ref as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dc,b,rows,P", [(6, 12, [75, 75, 80], 1), (3, 12, [60, 90], 1), (6, 4, [40, 47], 2), (3, 5, [60, 67], 3),
                                         (6, 19, [300, 307], 4), (6, 12, [75, 75, 75, 75], 3), (3, 19, [400], 8), (6, 7, [200, 30, 9], 5)])
def test_band_solver_matches_dense_solve(gpu_ctx, monkeypatch, dc, b, rows, P):
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", str(P))
    band, A, cp, rhs = R.random_band_system(rows, b, dc, seed=dc * 100 + b)
    X, info = ba.band_solve_probe(gpu_ctx, dc, cp, band, rhs)
    segs, seps = R.segment_table(cp, b, P)
    assert info["failed"] == 0 and info["separators"] == len(seps) and info["segments"] == len(segs)
    xr = np.linalg.solve(A, rhs.T).T
    assert np.abs(X - xr).max() <= 1e-12 * np.abs(xr).max()


@pytest.mark.parametrize("b,rows", [(21, [70, 95]), (22, [60, 23, 101]), (26, [120]), (26, [27, 26, 9]), (30, [64, 31, 150])])
@pytest.mark.parametrize("packed", ["1", "0"])
def test_wide_band_packed_window_matches_dense_solve(gpu_ctx, monkeypatch, b, rows, packed):
    """Half-widths beyond the square LDS window ring (6x6 blocks, 21..30): the packed-window factorisation of round 4 (band_kernels2p.h) + the back substitution with
    three task sets per lane, components longer and shorter than the window; SSFM_BAND_PACKED=0: the global-memory kernels they replace."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_BAND_PACKED", packed)
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", "1")
    band, A, cp, rhs = R.random_band_system(rows, b, 6, seed=600 + b)
    X, info = ba.band_solve_probe(gpu_ctx, 6, cp, band, rhs)
    assert info["failed"] == 0
    xr = np.linalg.solve(A, rhs.T).T
    assert np.abs(X - xr).max() <= 1e-12 * np.abs(xr).max()
    if packed == "1":
        band[rows[0] // 2, 0] -= np.eye(6) * 1e4               # a non-positive pivot raises the flag on this path too
        assert ba.band_solve_probe(gpu_ctx, 6, cp, band, rhs)[1]["failed"] == 1


@pytest.mark.parametrize("dc,b,rows,P", [(6, 5, [130], 4), (6, 5, [160, 171], 5), (6, 14, [420], 6), (3, 9, [300, 310], 7), (6, 7, [330], 10), (6, 3, [64], 4)])
def test_two_sided_separator_chain(gpu_ctx, monkeypatch, dc, b, rows, P):
    """Chains of three or more separators are eliminated from both ends by two workgroups that meet at the middle separator (band_sub.h 4b):
    odd and even chain lengths, partial last 16-tiles, several components, against numpy's dense solve."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", str(P))
    band, A, cp, rhs = R.random_band_system(rows, b, dc, seed=7 * dc + b + P)
    X, info = ba.band_solve_probe(gpu_ctx, dc, cp, band, rhs)
    assert info["failed"] == 0 and info["separators"] == (P - 1) * len(rows)
    xr = np.linalg.solve(A, rhs.T).T
    assert np.abs(X - xr).max() <= 1e-12 * np.abs(xr).max()
    for _ in range(3):                                      # the hand-over flags carry a launch number: repeated solves on fresh handles agree bit for bit
        X2, _ = ba.band_solve_probe(gpu_ctx, dc, cp, band, rhs)
        assert np.array_equal(X, X2)


def test_substructured_intermediates(gpu_ctx, monkeypatch):
    """Spikes Z, separator blocks D and right-hand sides t against the numpy statement of the same elimination."""
    from spherical_sfm_amd import ba
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", "3")
    dc, b = 6, 5
    band, A, cp, rhs = R.random_band_system([70, 64], b, dc, seed=9)
    X, info, Z, D, T = ba.band_solve_probe(gpu_ctx, dc, cp, band, rhs, dump=True)
    segs, seps = R.segment_table(cp, b, 3)
    Zr, Dr, Tr = R.substructure_intermediates(A, rhs, segs, seps, b, dc)
    for (r0, r1, re, has_left) in segs:
        if has_left:
            assert np.abs(Z[:, r0*dc:re*dc] - Zr[:, r0*dc:re*dc]).max() <= 1e-13
    for s in range(len(seps)):
        assert np.abs(np.tril(D[s]) - Dr[s]).max() <= 1e-12 * np.abs(Dr).max() and np.abs(T[s] - Tr[s]).max() <= 1e-13


def test_indefinite_band_raises_the_flag(gpu_ctx, monkeypatch):
    from spherical_sfm_amd import ba
    for P in (1, 2):
        monkeypatch.setenv("SSFM_BAND_SEGMENTS", str(P))
        band, A, cp, rhs = R.random_band_system([60], 4, 6, seed=2)
        band[30, 0] -= np.eye(6) * 1e3                     # a pivot inside a segment
        assert ba.band_solve_probe(gpu_ctx, 6, cp, band, rhs)[1]["failed"] == 1
    monkeypatch.setenv("SSFM_BAND_SEGMENTS", "2")
    band, A, cp, rhs = R.random_band_system([60], 4, 6, seed=2)
    segs, seps = R.segment_table(cp, 4, 2)
    band[seps[0][0] + 1, 0] -= np.eye(6) * 1e3             # a pivot inside the separator chain
    assert ba.band_solve_probe(gpu_ctx, 6, cp, band, rhs)[1]["failed"] == 1


@pytest.mark.parametrize("env", [{"SSFM_ASM_MFMA": "0"}, {"SSFM_CHAIN_DIAG_MFMA": "0"}, {"SSFM_CHAIN_PINGPONG": "0"}, {"SSFM_CHAIN_THREADS": "512"}, {"SSFM_CHAIN_MFMA": "0"}])
def test_substructured_solver_variants_agree_with_the_dense_solve(env):
    """The kernels the round-4 ones replaced stay selectable (VALU separator blocks with atomics, lane-per-row 16x16 factor-and-invert without prefetch, one
    triangle in LDS, eight waves, the VALU chain of round 1): each in its own process (the switches are read once), against numpy's dense solve --
    half-width 14 (84-row separators, two triangles fit) and 19 (114 rows: the LDS is full), two-sided chains."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import _band_ref as R; from spherical_sfm_amd import ba; ctx = ba.Context(0)\n"
            "for (b, rows, P) in ((14, [420], 6), (19, [300, 307], 4), (5, [160, 171], 5)):\n"
            "    import os; os.environ['SSFM_BAND_SEGMENTS'] = str(P)\n"
            "    band, A, cp, rhs = R.random_band_system(rows, b, 6, seed=31 + b)\n"
            "    X, info = ba.band_solve_probe(ctx, 6, cp, band, rhs)\n"
            "    xr = np.linalg.solve(A, rhs.T).T\n"
            "    assert info['failed'] == 0 and info['separators'] == (P - 1) * len(rows), info\n"
            "    err = np.abs(X - xr).max() / np.abs(xr).max(); assert err <= 1e-12, (b, err)\n"
            "print('VARIANT_OK')\n") % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "VARIANT_OK" in r.stdout, (env, r.stdout[-500:], r.stderr[-1500:])
