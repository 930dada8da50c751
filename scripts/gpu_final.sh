#!/bin/bash
# Final measurement pass (GPU box), to be run AFTER the last kernel commit so that profiles/ names HEAD's kernels.
#   bash scripts/gpu_final.sh <tag> pmc    rocprofv3 kernel stats + the four PMC passes at config 2, kernel stats at the configs[4] size, Retriangulate PMC
#   bash scripts/gpu_final.sh <tag> bench  bench.py (N = 1) + the 2-rank host-staged line + focal-free / spherical variants (record lines + the detail files)
# Summaries land in gpurun_out/; scripts/collect_final.py <tag> copies what is tracked into profiles/ (the PMC summary carries the digest of the kernel sources:
# bench.py prints pmc_stale = true when HEAD's differ).
TAG=${1:-r06f}; STAGE=${2:-bench}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=$GRAFT_REPO_ROOT/gpurun_out
first_csv() { find "$1" -name "$2" 2>/dev/null | head -1; }
if [ "$STAGE" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-scale-probe --no-side-paths"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -o ba -- $B --steps 5 --warmup 1 > $OUT/prof_${TAG}.log 2>&1
  F=$(first_csv $OUT/prof_${TAG} "*kernel_stats.csv"); [ -n "$F" ] && { cp "$F" $OUT/${TAG}_rocprofv3_kernel_stats.csv; head -12 "$F" | cut -c1-150; }
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1_${TAG} -o ba -- $B --steps 1 --warmup 0 > $OUT/pmc1_${TAG}.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2_${TAG} -o ba -- $B --steps 1 --warmup 0 > $OUT/pmc2_${TAG}.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3_${TAG} -o ba -- $B --steps 1 --warmup 0 > $OUT/pmc3_${TAG}.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc4_${TAG} -o ba -- $B --steps 1 --warmup 0 > $OUT/pmc4_${TAG}.log 2>&1
  for i in 1 2 3 4; do F=$(first_csv $OUT/pmc${i}_${TAG} "*counter_collection.csv"); [ -n "$F" ] && cp "$F" $OUT/pmc${i}_${TAG}/ba_counter_collection.csv 2>/dev/null; ls $OUT/pmc${i}_${TAG} | head -2; done
  CHECK=0 timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scale_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/scripts/dev/scale.py > $OUT/prof_scale_${TAG}.log 2>&1
  F=$(first_csv $OUT/prof_scale_${TAG} "*kernel_stats.csv"); [ -n "$F" ] && { cp "$F" $OUT/${TAG}_scale_rocprofv3_kernel_stats.csv; head -14 "$F" | cut -c1-150; }
  # round 5: the ring layout on irregular tracks and on the 4000-node pose graph (scripts/dev/ring.py: warm-up solve, five unprofiled solves, one with event brackets)
  for c in ragged14 ragged8 rot4000; do
    CHECK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${c}_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/scripts/dev/ring.py $c > $OUT/prof_${c}_${TAG}.log 2>&1
    F=$(first_csv $OUT/prof_${c}_${TAG} "*kernel_stats.csv"); [ -n "$F" ] && { cp "$F" $OUT/${TAG}_${c}_rocprofv3_kernel_stats.csv; head -8 "$F" | cut -c1-120; }
  done
  # Retriangulate (trace replay, 100k points x 6): kernel stats + the VALU / wait counters
  CHECK=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_retri_${TAG} -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 1 > $OUT/prof_retri_${TAG}.log 2>&1
  F=$(first_csv $OUT/prof_retri_${TAG} "*kernel_stats.csv"); [ -n "$F" ] && { cp "$F" $OUT/${TAG}_retri_rocprofv3_kernel_stats.csv; head -4 "$F" | cut -c1-150; }
  CHECK=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_retri_${TAG} -o retri -- python3 $GRAFT_REPO_ROOT/scripts/prof_retri.py 300 100000 6 1 > $OUT/pmc_retri_${TAG}.log 2>&1
  F=$(first_csv $OUT/pmc_retri_${TAG} "*counter_collection.csv"); [ -n "$F" ] && cp "$F" $OUT/${TAG}_retri_counter_collection.csv
  tail -2 $OUT/pmc_retri_${TAG}.log
else
  timeout 900 python bench.py --steps 20 --warmup 5 --detail $OUT/bench_${TAG}_detail.json > $OUT/bench_${TAG}.json 2> $OUT/bench_${TAG}.err; tail -c 400 $OUT/bench_${TAG}.json; tail -2 $OUT/bench_${TAG}.err
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1 --comm host --pairwise-pairs 400000 --detail $OUT/bench_${TAG}_2ranks_host_detail.json > $OUT/bench_${TAG}_2ranks_host.json 2> $OUT/bench_${TAG}_2ranks_host.err
  tail -c 300 $OUT/bench_${TAG}_2ranks_host.json
  timeout 300 python bench.py --steps 10 --warmup 2 --focal-free --no-side-paths --no-scale-probe --detail $OUT/bench_${TAG}_focalfree_detail.json > $OUT/bench_${TAG}_focalfree.json 2>> $OUT/bench_${TAG}.err
  timeout 300 python bench.py --steps 10 --warmup 2 --mode spherical --no-side-paths --no-scale-probe --detail $OUT/bench_${TAG}_spherical_detail.json > $OUT/bench_${TAG}_spherical.json 2>> $OUT/bench_${TAG}.err
  python3 -c "
import json
for v in ('focalfree','spherical'):
    d=json.loads(open('$OUT/bench_${TAG}_'+v+'.json').read().strip().splitlines()[-1]); print(v, d['value'], d['ms_per_step'], d.get('parity_vs_oracle'))"
fi
