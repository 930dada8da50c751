# GPU-box script: the three config-2 bench variants + the configs[4]-size probe.  Usage: bash scripts/gpu_bench_variants.sh <tag>
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err
python bench.py --steps 10 --warmup 2 --focal-free > gpurun_out/bench_${TAG}_focalfree.json 2>> gpurun_out/bench_${TAG}.err
python bench.py --steps 10 --warmup 2 --mode spherical > gpurun_out/bench_${TAG}_spherical.json 2>> gpurun_out/bench_${TAG}.err
python scripts/dev/scale.py > gpurun_out/scale_${TAG}.txt 2>&1
for f in gpurun_out/bench_${TAG}.json gpurun_out/bench_${TAG}_focalfree.json gpurun_out/bench_${TAG}_spherical.json; do
python - "$f" <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split('/')[-1], "obs/s %.3e  ms %.2f  cpu %.3e  x%.1f  parity %s  e2e cold %.1f ms x%.1f  warm %.1f ms x%.1f" % (
    d["value"], d["ms_per_step"], d["cpu_baseline"]["value"], d["speedup_vs_cpu_lm_loop"], d["parity_vs_oracle"]["max_rel_point"],
    1e3 * d["end_to_end_optimize"]["gpu_s"], d["end_to_end_optimize"]["speedup"], 1e3 * d["end_to_end_optimize"]["warm"]["gpu_s"], d["end_to_end_optimize"]["speedup_warm"]))
PY
done
tail -4 gpurun_out/scale_${TAG}.txt
