set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04d
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -30
( time timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
tail -c 600 $OUT/bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04d/bench.json').read().strip().splitlines()[-1])
print('value', b['value'], b['ms_per_step'])
print('roofline', json.dumps(b['roofline'])[:3000])
for k in ('host','lreg','kmeans','sparse','cpu_baseline'):
    print(k, json.dumps(b.get(k))[:1800])
PY
