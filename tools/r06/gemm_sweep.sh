#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
for shape in 2048,2048,2048 2304,2304,2304 3072,3072,3072 5000,5000,5000; do
  echo "== $shape"
  echo -n "default: "; python gemm_shapes.py $shape 2>&1 | grep -v amdgpu | tail -1
  for v in 6 7 8; do echo -n "variant $v: "; SP_GEMM_VARIANT=$v python gemm_shapes.py $shape 2>&1 | grep -v amdgpu | tail -1; done
  echo -n "SK=1: "; SP_GEMM_SK=1 python gemm_shapes.py $shape 2>&1 | grep -v amdgpu | tail -1
  echo -n "SK=0: "; SP_GEMM_SK=0 python gemm_shapes.py $shape 2>&1 | grep -v amdgpu | tail -1
done
