#!/bin/bash
# Final measurement pass of round 2 (GPU box): bench + kernel stats + PMC passes + configs[4]-size trace + the 2-rank host-staged line.
# Usage: bash scripts/gpu_final_r02.sh <tag>
TAG=${1:-r02d}
bash $GRAFT_REPO_ROOT/scripts/gpu_bench_profile.sh $TAG
cd $GRAFT_REPO_ROOT
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --comm host > gpurun_out/bench_${TAG}_2ranks_host.json 2> gpurun_out/bench_${TAG}_2ranks_host.err
tail -c 600 gpurun_out/bench_${TAG}_2ranks_host.json
python bench.py --steps 10 --warmup 2 --focal-free --no-side-paths > gpurun_out/bench_${TAG}_focalfree.json 2>> gpurun_out/bench_${TAG}.err
python bench.py --steps 10 --warmup 2 --mode spherical --no-side-paths > gpurun_out/bench_${TAG}_spherical.json 2>> gpurun_out/bench_${TAG}.err
