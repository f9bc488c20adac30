set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd $R
python tools/host_times.py > $OUT/host_times.txt 2>&1; tail -3 $OUT/host_times.txt
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -30
timeout 900 python bench.py --only hbm,lreg,kmeans > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -c 600 $OUT/bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04c/bench.json').read().strip().splitlines()[-1])
for k in ('hbm','lreg','kmeans'):
    print(k, json.dumps(b[k])[:1500])
PY
