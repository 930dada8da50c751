#!/bin/bash
# Builds libssfm_hip_ld<LD>.so for a few row strides of k_schur_gram's LDS half products (VERDICT r3 #5a: "measured, not guessed").  Run here; the .so files travel with gpurun.
cd $(dirname $0)/../spherical_sfm_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I/opt/rocm/include"
make -j4 ../libssfm_hip.so > /dev/null
for LD in ${@:-26 30 34 36}; do
  /opt/rocm/bin/hipcc $FLAGS -DSSFM_GRAM_LD=$LD -c -o build/ba_solver_ld$LD.o ba_solver.hip &
done
wait
for LD in ${@:-26 30 34 36}; do
  /opt/rocm/bin/hipcc $FLAGS -shared -o ../libssfm_hip_ld$LD.so build/ctx.o build/ba_solver_ld$LD.o build/rotavg_solver.o build/ransac.o build/lomsac.o build/retriangulate.o build/tracks.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
  ls -la ../libssfm_hip_ld$LD.so
done
