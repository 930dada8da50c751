"""Debug/measurement: GPU Retriangulate vs the oracle (which replays the reference's LO-MSAC with its random streams)."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from spherical_sfm_amd import ba, synth
from oracle import oracle

def corrupt(prob, frac, rng):
    """gross outliers: one observation of `frac` of the points moved by 30..80 px"""
    xy = prob.obs_xy.copy()
    pts = rng.choice(len(prob.points), int(frac * len(prob.points)), replace=False)
    first = {}
    for i, p in enumerate(prob.obs_pt):
        first.setdefault(int(p), []).append(i)
    for p in pts:
        i = first[int(p)][rng.integers(len(first[int(p)]))]
        ang = rng.uniform(0, 2 * np.pi); r = rng.uniform(30, 80)
        xy[i] += r * np.array([np.cos(ang), np.sin(ang)])
    prob.obs_xy = xy
    return pts

if __name__ == "__main__":
    Nc, Np, K = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (60, 2000, 6)
    rng = np.random.default_rng(5)
    prob = synth.make_circle(Nc, Np, K, rot_noise_deg=0.0, pixel_noise=0.5, seed=3)
    bad = corrupt(prob, 0.1, rng)
    ctx = ba.Context()
    ba.retriangulate(ctx, prob)
    t0 = time.time(); Xg, ng = ba.retriangulate(ctx, prob); tg = time.time() - t0
    t0 = time.time(); Xo, no = oracle.retriangulate(prob, 16); to = time.time() - t0
    zg = ~Xg.any(1); zo = ~Xo.any(1)
    both = ~zg & ~zo
    rel = np.linalg.norm(Xg - Xo, axis=1)[both] / np.linalg.norm(Xo[both], axis=1)
    print(f"gpu {tg*1e3:.1f} ms  oracle {to:.2f} s   zero: gpu {zg.sum()} oracle {zo.sum()} mismatch {(zg != zo).sum()}")
    print("inlier count mismatch", (ng != no).sum(), " hist gpu", np.bincount(ng), "oracle", np.bincount(no))
    print("rel diff quantiles 50/90/99/100:", np.quantile(rel, [0.5, 0.9, 0.99, 1.0]))
    gtX = prob.gt_points
    eg = np.linalg.norm(Xg - gtX, axis=1)[both] / np.linalg.norm(gtX[both], axis=1)
    eo = np.linalg.norm(Xo - gtX, axis=1)[both] / np.linalg.norm(gtX[both], axis=1)
    print("vs GT median gpu/oracle", np.median(eg), np.median(eo), " max", eg.max(), eo.max())
    def msac(X):
        from spherical_sfm_amd.synth import so3exp
        R = so3exp(prob.cameras[:, 3:]); t = prob.cameras[:, :3]
        pc = np.einsum('nij,nj->ni', R[prob.obs_cam], X[prob.obs_pt]) + t[prob.obs_cam]
        e = ((prob.focal * pc[:, :2] / pc[:, 2:3] - prob.obs_xy) ** 2).sum(1)
        e = np.where(pc[:, 2] < 0, np.inf, e)
        return np.bincount(prob.obs_pt, np.minimum(e, 4.0), len(X))
    sg, so = msac(Xg), msac(Xo)
    d = (sg - so)[both]
    print("score gpu-oracle: min %.3e max %.3e  n(gpu better by >1e-6) %d  n(oracle better by >1e-6) %d" % (d.min(), d.max(), (d < -1e-6).sum(), (d > 1e-6).sum()))
    same = (ng == no)[both]
    print("rel diff where inlier counts agree: max %.3e  q999 %.3e" % (rel[same].max(), np.quantile(rel[same], 0.999)))
    w = np.argsort(-rel)[:5]
    idx = np.nonzero(both)[0][w]
    for i in idx: print(i, Xg[i], Xo[i], ng[i], no[i], i in set(bad.tolist()))
