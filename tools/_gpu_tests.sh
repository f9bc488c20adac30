set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04e
mkdir -p $OUT
cd $R
python tools/_exp/first_call.py | cut -c1-110 | head -4
timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | head -30
timeout 900 python bench.py --only hbm,host,lreg,sparse > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
b=json.loads(open('gpurun_out/r04e/bench.json').read().strip().splitlines()[-1])
h=b['hbm']
print({k:v for k,v in h.items() if 'chain' in k}, h['stream_copy_GBps'], h['frac_of_measured_copy']['map_5op_chain_first_call_GBps'])
print(b['host']); print(b['lreg']['ms_per_step'], b['sparse']['spmv_ms'])
PY
