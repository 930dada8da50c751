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
other scripts (scripts/): gpu_final.sh + collect_final.py + pmc_summary.py (the final measurement pass: rocprofv3 stats, PMC passes, bench lines -> profiles/), gpu_retry.sh (retry a
gpurun call while the pod's slots are busy), bench_pairwise.py (configs[3] pairwise RANSAC), prof_gram_*.py / prof_irregular.py / prof_retri.py / prof_rot.py (single-workload
profiling drivers named in DESIGN.md and the profiles/ notes), soak_*.py (randomised soaks: band solver vs numpy, repeated BA solves, Gram fuzz, Retriangulate trace),
build_gram_ld_variants.sh (library variants for SSFM_LIB_PATH), r06/ab.sh (same-run A/B of environment settings at config 2), r06/snode_stamps.py (phase stamps of k_snode_solve),
r06/reflow_md.py (Markdown reflow to 160 columns), r06/ab_mode.sh + ragged_ab.sh + gbs_ab.sh (the same for bench.py flags / the ragged problems / the configs[4] size),
r06/build.sh (library + lab build from anywhere), r06/k8_probe.py (the Gram kernels on 109k points x 8), kernel_resources.py (registers / scratch / spills of every kernel in the
built library, from the code objects' metadata: no GPU; tests/test_kernel_resources_cpu.py), lab/ (kernel labs and A/B scripts of rounds 2-5: chol_lab, ldl16_lab, syrk_lab, point_lab, ab_*.sh)
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
