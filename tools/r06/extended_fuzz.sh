#!/bin/bash
# Extended differential fuzz of the final tree on one GPU box: output kept as profiles/r06_extended_fuzz.txt.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_extended_fuzz.txt
: > $OUT
echo "tree $(python -c 'from spartan_amd import _hip; print(_hip.source_sha())')" >> $OUT
echo "== tests/test_fuzz_gpu.py, SPARTAN_FUZZ_N=3000 (per worker count)" >> $OUT
SPARTAN_FUZZ_N=3000 timeout 1500 python -m pytest tests/test_fuzz_gpu.py tests/test_fuzz_builders.py -q 2>&1 | tail -3 >> $OUT
cd tools
for seed in 11 12 13 14 15; do
  echo "== fuzz_kmeans.py seed $seed" >> ../$OUT
  timeout 900 python fuzz_kmeans.py $seed 2>&1 | tail -2 >> ../$OUT
done
for seed in 21 22 23; do
  echo "== fuzz_gemm.py seed $seed" >> ../$OUT
  timeout 900 python fuzz_gemm.py $seed 2>&1 | tail -2 >> ../$OUT
done
cd ..
echo "== sparse: scipy fuzz + goldens" >> $OUT
timeout 900 python -m pytest tests -m gpu -q -k "sparse or spmv or spmm" 2>&1 | tail -2 >> $OUT
cat $OUT
