#!/bin/bash
# Round-2 bench + profile pass.  Usage: bash scripts/gpu_bench_r02.sh <tag>
TAG=${1:-r02}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 1500 gpurun_out/bench_${TAG}.json; tail -3 gpurun_out/bench_${TAG}.err
# the N > 1 code path of bench.py on this 1-GPU box: two ranks sharing GPU 0, reductions through the host hook (NOT a performance number)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --comm host > gpurun_out/bench_${TAG}_2ranks_host.json 2> gpurun_out/bench_${TAG}_2ranks_host.err
tail -c 1200 gpurun_out/bench_${TAG}_2ranks_host.json; tail -3 gpurun_out/bench_${TAG}_2ranks_host.err
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-scale-probe > $OUT/prof_${TAG}.log 2>&1
head -16 $(find $OUT/prof_${TAG} -name "*kernel_stats.csv" | head -1) | cut -c1-150
CHECK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scale_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/scripts/dev/scale.py > $OUT/prof_scale_${TAG}.log 2>&1
head -14 $(find $OUT/prof_scale_${TAG} -name "*kernel_stats.csv" | head -1) | cut -c1-150
