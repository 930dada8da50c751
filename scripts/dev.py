"""Developer probes, one entry point:  python scripts/dev.py <name> [args...]   (runs scripts/dev/<name>.py; GPU box unless noted)

timing
  ba              config-2 bundle adjustment in the four modes (spherical / general x focal fixed / free): ms per solve + parity vs the oracle
  warm            repeated ssfm_ba_solve on one structure (plan cache hit)
  cold            first call on a structure: plan / upload / solve / download (SSFM_PLAN_TIMING=1 prints the planner's stages)
  e2e             end-to-end Optimize() per call
  plan            host planner only (no GPU)
  scale           BASELINE configs[4] size (4000 cameras / 1.5 M points / 12 M observations) on one GPU; CHECK=0 skips the oracle
  rotavg, rot_time, rot_large, rot_cut   pose-graph solves: config-2 size, timing loop, 2000 / 4000 nodes, forced segment counts
  rot_diverge     how fast the device and the oracle part on the 2000 / 4000-node pose graphs (relative cost / angle difference after k iterations)
parity / debugging
  hard            BA starts that make the LM reject steps
  sub [dc b rows P]   the band solver probe against numpy, stage by stage
  ransac, trace, trace_mismatch, lsq_replay   pairwise LO-MSAC: batch vs oracle, where a reference-trace run leaves the oracle's
  retri, tri_bits Retriangulate vs the oracle; which bits of DLT / score / least squares differ
other scripts of round 4 (scripts/): soak_band_solver.py (randomised soak of the substructured solver against numpy), sweep_segments_r04.sh (SSFM_BAND_SEGMENTS sweep at the
configs[4] size and on the large pose graphs), prof_rot_rocprof.sh (rocprofv3 kernel stats of optimize_rotations), gpu_final_r04.sh + collect_final_r04.py (the final measurement pass),
lab/ldl16_lab.hip (16x16 factor-and-invert: lane per row vs matrix cores), lab/syrk_lab.hip (the chain's rank-Q update alone), lab/ab_*.sh (same-run A/B of an environment switch)
"""
import os
import runpy
import sys

if __name__ == "__main__":
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dev")
    if len(sys.argv) < 2 or not os.path.exists(os.path.join(here, sys.argv[1] + ".py")):
        print(__doc__); print("available:", " ".join(sorted(f[:-3] for f in os.listdir(here) if f.endswith(".py")))); sys.exit(1)
    name = sys.argv[1]; sys.argv = [os.path.join(here, name + ".py")] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")
