"""debug: where do the device DLT / score / least-squares bits differ from the oracle's?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from spherical_sfm_amd import ba, synth
from oracle import oracle as O
prob = synth.make_circle(60, 400, 6, rot_noise_deg=0.0, pixel_noise=0.5, seed=3)
synth.corrupt_observations(prob, 0.1, seed=5)
ctx = ba.Context(0)
K = 6
tp, lists = [], []
for j in range(0, 400):
    for a in range(K):
        for b in range(K):
            if a != b:
                tp.append(j); lists.append([a, b])
g = ba.tri_probe(ctx, prob, 0, tp, lists); o = O.tri_probe(prob, 0, tp, lists)
bad = ~((g == o) | (np.isnan(g) & np.isnan(o))).all(1)
print("DLT mismatches", bad.sum(), "of", len(bad))
tpa = np.array(tp); la = np.array(lists)
print("points with mismatches", np.unique(tpa[bad])[:40])
cams = prob.obs_cam.reshape(400, K)
badcams = {}
for t in np.nonzero(bad)[0]:
    for k in la[t]:
        c = cams[tpa[t], k]; badcams[c] = badcams.get(c, 0) + 1
print("cameras in mismatching samples", sorted(badcams.items()))
allc = {}
for t in range(len(tp)):
    for k in la[t]:
        c = cams[tpa[t], k]; allc[c] = allc.get(c, 0) + 1
print("fraction bad per camera", {c: round(badcams.get(c, 0) / allc[c], 2) for c in sorted(allc)})
rel = np.abs(g[bad, :3] - o[bad, :3]).max(1) / np.abs(o[bad, :3]).max(1)
print("relative differences: median %.2e max %.2e" % (np.median(rel), rel.max()))
for t in np.nonzero(bad)[0][:3]:
    print(tp[t], lists[t], g[t], o[t], "cams", cams[tp[t], lists[t]], prob.cameras[cams[tp[t], lists[t]]])
