#!/bin/bash
# usage: build_spmv_variant.sh NAME -DBSP_ABLATE=1 ...   -> tools/r06/libspartan_hip_NAME.so
set -e
NAME=$1; shift
cd /root/repo/spartan_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable "$@" -c spmv_blocked.hip -o ../../tools/r06/spmv_$NAME.o
OBJS=$(ls map.o reduce.o argreduce.o update.o gemm.o gemm_f64.o sp_jit.o random.o sparse.o kmeans.o rowdot.o tiling.o runtime.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/r06/libspartan_hip_$NAME.so $OBJS ../../tools/r06/spmv_$NAME.o -ldl
echo built $NAME
