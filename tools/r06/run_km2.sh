#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  echo -n "$v: "; SP_KM_COOP=0 SPARTAN_HIP_LIB=$PWD/tools/r06/libspartan_hip_$v.so timeout 300 python tools/km_first.py 2>&1 | tail -1
done
done
