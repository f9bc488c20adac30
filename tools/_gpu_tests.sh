set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04a
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -70
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench.err
timeout 600 python bench.py --gpus 2 --size 4096 --steps 3 --warmup 1 > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench2 rc=$?"
tail -c 1500 $OUT/bench2.err
head -c 3000 $OUT/bench2.json
