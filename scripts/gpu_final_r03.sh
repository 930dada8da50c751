#!/bin/bash
# Final measurement pass of round 3 (GPU box).  Usage: bash scripts/gpu_final_r03.sh <tag> <stage>
#   stage pmc   : rocprofv3 kernel stats + the three PMC passes at config 2, kernel stats at the configs[4] size
#   stage bench : bench.py (N = 1) + the 2-rank host-staged line + focal-free / spherical variants
TAG=${1:-r03f}; STAGE=${2:-bench}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
first_csv() { find "$1" -name "$2" 2>/dev/null | head -1; }
if [ "$STAGE" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-scale-probe > $OUT/prof_${TAG}.log 2>&1
  F=$(first_csv $OUT/prof_${TAG} "*kernel_stats.csv"); [ -n "$F" ] && { cp "$F" $OUT/${TAG}_rocprofv3_kernel_stats.csv; head -14 "$F" | cut -c1-140; }
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/pmc1_${TAG}.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/pmc2_${TAG}.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/pmc3_${TAG}.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc4_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/pmc4_${TAG}.log 2>&1
  F=$(first_csv $OUT/pmc4_${TAG} "*counter_collection.csv"); [ -n "$F" ] && cp "$F" $OUT/pmc4_${TAG}/ba_counter_collection.csv 2>/dev/null; tail -3 $OUT/pmc4_${TAG}.log
  for i in 1 2 3; do F=$(first_csv $OUT/pmc${i}_${TAG} "*counter_collection.csv"); [ -n "$F" ] && cp "$F" $OUT/pmc${i}_${TAG}/ba_counter_collection.csv 2>/dev/null; ls $OUT/pmc${i}_${TAG} | head -3; done
  CHECK=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scale_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/scripts/dev/scale.py > $OUT/prof_scale_${TAG}.log 2>&1
  F=$(first_csv $OUT/prof_scale_${TAG} "*kernel_stats.csv"); [ -n "$F" ] && { cp "$F" $OUT/${TAG}_scale_rocprofv3_kernel_stats.csv; head -16 "$F" | cut -c1-140; }
else
  timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err; tail -c 600 $OUT/bench_${TAG}.json; tail -2 $OUT/bench_${TAG}.err
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --comm host --pairwise-pairs 400000 > $OUT/bench_${TAG}_2ranks_host.json 2> $OUT/bench_${TAG}_2ranks_host.err
  tail -c 400 $OUT/bench_${TAG}_2ranks_host.json
  timeout 300 python bench.py --steps 10 --warmup 2 --focal-free --no-side-paths --no-scale-probe > $OUT/bench_${TAG}_focalfree.json 2>> $OUT/bench_${TAG}.err
  timeout 300 python bench.py --steps 10 --warmup 2 --mode spherical --no-side-paths --no-scale-probe > $OUT/bench_${TAG}_spherical.json 2>> $OUT/bench_${TAG}.err
  python3 -c "
import json
for v in ('focalfree','spherical'):
    d=json.load(open('$OUT/bench_${TAG}_'+v+'.json')); print(v, d['value'], d['ms_per_step'], d.get('parity_vs_oracle'))"
fi
