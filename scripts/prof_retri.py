"""Retriangulate at config-2 size under rocprofv3 / plain timing: trace mode (default) and the enumerating kernel (ssfm_retriangulate_mode).
usage: python scripts/prof_retri.py [Nc Np K] [repeats]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spherical_sfm_amd import ba, synth
Nc, Np, K = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (300, 100000, 6)
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
prob = synth.make_circle(Nc, Np, K, rot_noise_deg=0.0, pixel_noise=0.5, seed=3)
synth.corrupt_observations(prob, 0.1, seed=5)
ctx = ba.Context(0)
for mode in ("trace", "enumerate"):
    m = ba.RETRI_MODE_ENUMERATE if mode == "enumerate" else ba.RETRI_MODE_TRACE
    ba.retriangulate(ctx, prob, mode=m)
    ts = []
    for _ in range(rep):
        t0 = time.perf_counter(); X, n = ba.retriangulate(ctx, prob, mode=m); ts.append(time.perf_counter() - t0)
    print(f"{mode}: {Np} points x {K} observations: {min(ts) * 1e3:.1f} ms end to end (host lists + upload + kernel + download), nonzero {X.any(1).sum()}")
if os.environ.get("CHECK", "1") != "0":
    from oracle import oracle as O
    t0 = time.perf_counter(); O.retriangulate(prob, 16); print(f"oracle ({os.cpu_count()} host cores, 16 threads): {(time.perf_counter() - t0) * 1e3:.0f} ms")
ctx.close()
