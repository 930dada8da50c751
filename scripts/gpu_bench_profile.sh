# GPU-box script: bench + rocprofv3 kernel stats + PMC passes.  Usage: bash scripts/gpu_bench_profile.sh <tag>
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 2500 gpurun_out/bench_${TAG}.json; tail -3 gpurun_out/bench_${TAG}.err
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-scale-probe > $OUT/prof_${TAG}.log 2>&1
find $OUT/prof_${TAG} -name "*stats*" | head; head -30 $(find $OUT/prof_${TAG} -name "*kernel_stats.csv" | head -1)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/pmc1_${TAG}.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc2_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/pmc2_${TAG}.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc3_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-scale-probe > $OUT/pmc3_${TAG}.log 2>&1
ls $OUT/pmc1_${TAG} $OUT/pmc2_${TAG} | head; tail -3 $OUT/pmc1_${TAG}.log
# configs[4]-size probe under the kernel trace (substructured factorisation)
CHECK=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_scale_${TAG} -o ba -- python3 $GRAFT_REPO_ROOT/scripts/dev/scale.py > $OUT/prof_scale_${TAG}.log 2>&1
head -24 $(find $OUT/prof_scale_${TAG} -name "*kernel_stats.csv" | head -1)
