import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from scipy.spatial.transform import Rotation
from spherical_sfm_amd import synth, ba, ransac
from oracle import oracle as O
ctx = ba.Context(0)
def frob_err(E, Es):
    a = E / np.linalg.norm(E); b = Es / np.linalg.norm(Es); return min(np.linalg.norm(a - b), np.linalg.norm(a + b))
def rot_err(R, Rs): return np.linalg.norm(Rotation.from_matrix(Rs @ R.T).as_rotvec())
u, v, R, E, _ = synth.make_relative_pose_problem(40, seed=3, noise=1e-3)
rng = np.random.default_rng(0)
samples = np.array([rng.choice(40, 3, replace=False) for _ in range(200)], np.int32)
got = ransac.solver_probe(ctx, u, v, samples)
errs=[]
for s, Es in zip(samples, got):
    ref = O.spherical_solver(u, v, s)
    for e in Es: errs.append(min(frob_err(e, r) for r in ref))
errs=np.array(errs); print('solver errs: n', len(errs), 'median', np.median(errs), 'q95', np.quantile(errs,.95), 'q99', np.quantile(errs,.99), 'max', errs.max(), '#>1e-6', (errs>1e-6).sum())
thr=(2/600)**2
probs=[synth.make_relative_pose_problem(150, seed=100+k, noise=1/600, outlier_frac=0.3, rotation_deg=5+(k%30)) for k in range(48)]
for H in (256, 1024, 4096):
    out = ransac.estimate_pairs(ctx, [(p[0],p[1]) for p in probs], thr, num_hypotheses=H, min_num_inliers=20)
    agree=[];ang=[];gt_g=[];gt_o=[];sc=[]
    for k,(u,v,R,E,inl) in enumerate(probs):
        o=O.ransac_pair(u,v,thr,min_num_inliers=20)
        agree.append((out['inliers'][k]==o['inliers']).mean()); ang.append(rot_err(o['R'],out['R'][k])); gt_g.append(rot_err(R,out['R'][k])); gt_o.append(rot_err(R,o['R'])); sc.append((out['scores'][k], o['score']))
    agree=np.array(agree); ang=np.array(ang)
    print('H',H,'agree mean %.4f min %.4f | ang median %.2e max %.2e | err vs GT gpu %.2e oracle %.2e | gpu score <= oracle score in %d/48' % (agree.mean(), agree.min(), np.median(ang), ang.max(), np.mean(gt_g), np.mean(gt_o), sum(a<=b*(1+1e-9) for a,b in sc)))
import time
big=[synth.make_relative_pose_problem(500, seed=k, noise=1/600, outlier_frac=0.3, rotation_deg=5+(k%30)) for k in range(64)]
pairs=[(p[0],p[1]) for p in big]*32
t=time.time(); out=ransac.estimate_pairs(ctx, pairs, thr, num_hypotheses=1024); dt=time.time()-t
print('%d pairs x 500 corr x 1024 hyp: %.3f s -> %.1f pairs/s' % (len(pairs), dt, len(pairs)/dt))
