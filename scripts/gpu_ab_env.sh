#!/bin/bash
# GPU box: A/B of an environment knob on the bench's per-kernel times (config 2 and the configs[4] size).  Usage: bash scripts/gpu_ab_env.sh VAR=value [VAR2=value2 ...]
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for setting in "BASE=1" "$@"; do
  env $setting timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
  python - "$setting" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d["kernels"]; c4 = d.get("roofline_configs4", {}).get("all_kernels_avg_us", {})
print(sys.argv[1], "| config2: ms/step %.3f" % d["ms_per_step"], " ".join(f"{n}={v['avg_us']:.1f}" for n, v in k.items() if n in ("k_schur_pairs2", "k_schur_gram", "k_cam_sums2", "k_point_lin", "k_point_backsub", "k_gram_backsub", "k_band_chol_v2", "k_band_back_v2")),
      "| configs4 size: ms/solve %.2f" % d.get("scale_probe_configs4_size_one_gpu", {}).get("ms_per_solve", float("nan")), " ".join(f"{n}={v:.0f}" for n, v in c4.items() if n in ("k_schur_pairs2", "k_schur_gram", "k_cam_sums2", "k_point_lin", "k_point_backsub", "k_gram_backsub", "k_band_chol_v2", "k_sub_sep_chain")))
PY
done
done
