cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/drv
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/drv/km -o km -- python $R/tools/driver_profile.py kmeans > $R/gpurun_out/drv/kmeans_prof.txt 2>&1
cd $R
python - <<'PY'
import glob, sys
sys.path.insert(0, 'tools')
import roofline
rows = roofline.load_trace(glob.glob('gpurun_out/drv/km/**/*kernel_trace.csv', recursive=True))
# the last 20 fit() iterations = the last 20 launches of the first-pass assign kernel; take everything after the 21st-from-last
idx = [i for i, r in enumerate(rows) if r['name'].startswith('sp_nearest_nt_kernel<true, false, false>')]
start = idx[-20]
sel = rows[start:]
tot = {}
for r in sel:
  t = tot.setdefault(r['name'], [0, 0.0]); t[0] += 1; t[1] += (r['end'] - r['start']) / 1e3
span = (sel[-1]['end'] - sel[0]['start']) / 1e3
print('20 iterations: span %.1f us per iteration, kernel sum %.1f us per iteration' % (span / 20, sum(v[1] for v in tot.values()) / 20))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
  print('%9.1f us/iter %5.1f calls/iter  %s' % (v[1] / 20, v[0] / 20.0, k[:100]))
PY
