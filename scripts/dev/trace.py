import sys, time, numpy as np
sys.path.insert(0, ".")
from spherical_sfm_amd import synth, ba, ransac
from oracle import oracle as O
from scipy.spatial.transform import Rotation
ctx = ba.Context(0)
THR = (2 / 600) ** 2
def frob_err(E, Es):
    a = E / np.linalg.norm(E); b = Es / np.linalg.norm(Es); return min(np.linalg.norm(a - b), np.linalg.norm(a + b))
def rot_err(R, Rs): return np.linalg.norm(Rotation.from_matrix(Rs @ R.T).as_rotvec())
probs = [synth.make_relative_pose_problem(500, seed=100 + k, noise=1 / 600, outlier_frac=0.3, rotation_deg=5 + (k % 30)) for k in range(64)]
out = ransac.estimate_pairs(ctx, [(p[0], p[1]) for p in probs], THR, min_num_inliers=20)
fe, re, same = [], [], []
for k, (u, v, R, E, inl) in enumerate(probs):
    o = O.lomsac_pair(u, v, THR, min_num_inliers=20)
    fe.append(frob_err(out["E"][k], o["E"])); re.append(rot_err(out["R"][k], o["R"]))
    same.append((out["iterations"][k] == o["iterations"], out["lo_runs"][k] == o["lo_runs"], (out["inliers"][k] == o["inliers"]).all()))
print("E err max/median", max(fe), np.median(fe), "R err max/median", max(re), np.median(re), "same", np.array(same).mean(axis=0))
print("iterations", np.bincount(out["iterations"])[100:].nonzero()[0][:10] + 100, "lo", np.bincount(out["lo_runs"]))
# timing: 20k pairs x 500
ptr = np.arange(0, 20001, dtype=np.int32) * 500
U = np.concatenate([p[0] for p in probs] * 313)[:20000 * 500]; V = np.concatenate([p[1] for p in probs] * 313)[:20000 * 500]
for mode in (1, 0):
    for rep in range(2):
        t = time.time(); o = ransac.estimate_flat(ctx, ptr, U, V, THR, min_num_inliers=20, mode=mode); dt = time.time() - t
    print(f"mode {mode}: 20000 pairs x 500 in {dt*1e3:.1f} ms = {20000/dt:.0f} pairs/s (end to end, incl. H2D of {U.nbytes*2/1e6:.0f} MB)")
ptr2, U2, V2, pairs, Rgt = synth.make_circle_pairs(200, 3000)
t = time.time(); o = ransac.estimate_flat(ctx, ptr2, U2, V2, THR, min_num_inliers=20); print("circle 200:", time.time() - t, "s", (o["num_inliers"] > 20).sum(), "accepted")
