"""Timing study of k_schur_gram: SSFM_GRAM_STAMPS=1 makes the first launch of a solve record per-task phase stamps (100 MHz clock) and print their means.
Usage: SSFM_GRAM_STAMPS=1 [SSFM_GRAM_PTS=n] python scripts/prof_gram_stamps.py [small|big]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from spherical_sfm_amd import ba, synth

which = sys.argv[1] if len(sys.argv) > 1 else "big"
ctx = ba.Context(0)
if which == "big":
    prob = synth.make_circle(4000, 1500000, 8, spherical=False, focal_fixed=True)
elif which == "config2":
    prob = synth.make_circle(300, 100000, 6, spherical=False, focal_fixed=True)
else:
    prob = synth.make_circle(600, 100000, 6, spherical=False, focal_fixed=True)
_, _, _, s = ba.optimize(ctx, prob)
print(which, s["iterations"], s["final_cost"])
ctx.close()
