#!/bin/bash
set -e
NAME=$1; shift
cd /root/repo/spartan_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable "$@" -c gemm.hip -o ../../tools/r06/gemm_$NAME.o
OBJS=$(ls map.o reduce.o argreduce.o update.o gemm_f64.o sp_jit.o kmeans.o random.o sparse.o spmv_blocked.o rowdot.o tiling.o runtime.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/r06/libspartan_hip_$NAME.so $OBJS ../../tools/r06/gemm_$NAME.o -ldl
echo built $NAME
