#!/bin/bash
cd $GRAFT_REPO_ROOT/tools
for rep in 1 2; do
echo -n "base: "; timeout 120 python spmv_time.py 2>&1 | grep "blocked)"
for v in "$@"; do
  echo -n "$v: "; SPARTAN_HIP_LIB=$GRAFT_REPO_ROOT/tools/r06/libspartan_hip_$v.so timeout 120 python spmv_time.py 2>&1 | grep "blocked)"
done
done
