"""Retriangulate kernel-path timing at a few sizes and register budgets (run on the GPU box): python scripts/prof_retri_sizes.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from spherical_sfm_amd import ba, synth
ctx = ba.Context(0)
for Nc, Np, stride in [(300, 100000, 4), (500, 170000, 7), (300, 131000, 4)]:
    p = synth.make_circle(Nc, Np, 6, rot_noise_deg=0.0, pixel_noise=0.5)
    for w in ["2", "3", None]:
        if w is None: os.environ.pop("SSFM_RETRI_WAVES", None)
        else: os.environ["SSFM_RETRI_WAVES"] = w
        ba.retriangulate(ctx, p)
        t = time.perf_counter(); X, nin = ba.retriangulate(ctx, p); dt = time.perf_counter() - t
        print(f"{Nc} x {Np}: waves/SIMD {w or 'auto'}: {1e3 * dt:.1f} ms end to end, zeroed {int((~X.any(1)).sum())}")
