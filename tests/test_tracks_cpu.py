"""Bit-exact parity of the track builder with the pure-Python oracle; host-only, runs on CPU."""
import numpy as np
import pytest

from oracle import tracks_oracle
from spherical_sfm_amd import tracks


def random_case(rng, K, nfeat, density, chain=True):
    feats = [rng.uniform(0, 1000, size=(nfeat, 2)) for _ in range(K)]
    pairs = [(i, i + d) for d in (1, 2, 3) for i in range(K - d)] if chain else [(i, j) for i in range(K) for j in range(i + 1, K)]
    ims = []
    for a, b in pairs:
        n = rng.integers(0, int(density * nfeat) + 1)
        f0 = rng.choice(nfeat, n, replace=False); f1 = rng.choice(nfeat, n, replace=False)
        ims.append((a, b, list(zip(f0.tolist(), f1.tolist()))))
    return feats, ims


@pytest.mark.parametrize("merge", [True, False])
@pytest.mark.parametrize("seed", range(6))
def test_tracks_bit_exact(seed, merge):
    rng = np.random.default_rng(seed)
    feats, ims = random_case(rng, K=int(rng.integers(3, 9)), nfeat=int(rng.integers(5, 40)), density=0.6, chain=bool(seed % 2))
    got = tracks.build_tracks(feats, ims, 500.0, 400.0, merge)
    ref = tracks_oracle.build_tracks(feats, ims, 500.0, 400.0, merge)
    assert [t.tolist() for t in got["tracks"]] == ref["tracks"]                    # bit-exact track indices (north_star)
    assert got["num_points"] == ref["num_points"] and got["alive"].tolist() == ref["alive"]
    assert list(zip(got["obs_cam"].tolist(), got["obs_pt"].tolist())) == [(c, p) for c, p, _ in ref["obs"]]
    assert np.array_equal(got["obs_xy"], np.array([xy for _, _, xy in ref["obs"]]).reshape(-1, 2))    # same float ops -> identical bits


def test_merge_semantics_small_example():
    feats = [np.array([[10., 10.], [20., 20.]]), np.array([[11., 11.], [21., 21.]]), np.array([[12., 12.]])]
    # 0-1 creates track 0 (f0-f0); 1-2 on another feature creates track 1 (f1-f0); 0-2 links feature 0 of kf0 (track 0) with kf2's (track 1)
    ims = [(0, 1, [(0, 0)]), (1, 2, [(1, 0)]), (0, 2, [(0, 0)])]
    m = tracks.build_tracks(feats, ims, merge=True)
    assert [t.tolist() for t in m["tracks"]] == [[0, -1], [0, 0], [0]] and m["alive"].tolist() == [True, False]
    assert sorted(zip(m["obs_cam"].tolist(), m["obs_pt"].tolist())) == [(0, 0), (1, 0), (2, 0)]
    # camera 1 saw both tracks: MergePoint copies the removed point's observation over the kept one (src/sfm.cpp:136)
    assert m["obs_xy"][list(m["obs_cam"]).index(1)].tolist() == [21., 21.]
    n = tracks.build_tracks(feats, ims, merge=False)
    assert [t.tolist() for t in n["tracks"]] == [[0, -1], [0, 1], [1]] and n["alive"].tolist() == [True, True]
    assert sorted(zip(n["obs_cam"].tolist(), n["obs_pt"].tolist())) == [(0, 0), (1, 0), (1, 1), (2, 1)]


def test_empty_and_unmatched():
    feats = [np.zeros((3, 2)), np.zeros((0, 2)), np.zeros((2, 2))]
    out = tracks.build_tracks(feats, [(0, 2, [])], merge=True)
    assert [t.tolist() for t in out["tracks"]] == [[-1, -1, -1], [], [-1, -1]] and out["num_points"] == 0 and len(out["obs_cam"]) == 0
