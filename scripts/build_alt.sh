#!/bin/bash
# Development aid: a second libcholamd.so with extra -D flags (A/B runs in one gpurun call):
#   scripts/build_alt.sh NAME [-DFOO=1 ...]   ->  cholesky_amd/lib/alt_NAME/libcholamd.so   (select with CHOLAMD_LIB=...)
# flags apply to chol_kernels.hip, chol_kernels_f32.hip and chol_schedule.c
set -e
name=$1; shift
out=cholesky_amd/lib/alt_$name
mkdir -p $out
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -Iinclude -Icholesky_amd/csrc -Wall -mllvm -pragma-unroll-threshold=100000 "$@" -c cholesky_amd/csrc/chol_kernels.hip -o $out/chol_kernels.o
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -Iinclude -Icholesky_amd/csrc -Wall "$@" -c cholesky_amd/csrc/chol_kernels_f32.hip -o $out/chol_kernels_f32.o
gcc -O2 -fPIC -Wall -Wextra -std=gnu11 -Iinclude -Icholesky_amd/csrc "$@" -c cholesky_amd/csrc/chol_schedule.c -o $out/chol_schedule.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libcholamd.so cholesky_amd/lib/chol_ingest.o cholesky_amd/lib/chol_symbolic.o $out/chol_schedule.o \
  cholesky_amd/lib/chol_generate.o $out/chol_kernels.o $out/chol_kernels_f32.o cholesky_amd/lib/chol_api.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $out/libcholamd.so
