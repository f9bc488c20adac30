set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03b
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -n 4 --maxfail=60 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -70
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 1500 $OUT/bench.err
head -c 6000 $OUT/bench.json
