set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04f
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -30
timeout 900 python bench.py --only kmeans > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 400 $OUT/bench.err
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04f/bench.json').read().strip().splitlines()[-1])
print(b['kmeans'])
PY
timeout 600 python bench.py --gpus 2 --size 2048 --steps 2 --warmup 1 --only kmeans_dist 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b.get('kmeans_dist'), b.get('error'))"
