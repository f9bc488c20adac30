set -x
cd /root/repo
timeout 900 python -m pytest tests/test_examples.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2 3; do timeout 300 python bench.py --only kmeans 2>&1 | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l); print({k:v for k,v in d.get('sections',d).items() if 'kmeans' in k})
"; done
