cd /root/repo
for i in 1 2 3; do timeout 300 python bench.py --only kmeans 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
  if l.startswith('{'):
    d=json.loads(l)
    def find(o):
      if isinstance(o,dict):
        if 'assign_ms' in o: return o
        for v in o.values():
          r=find(v)
          if r: return r
    h=find(d); print({k:h[k] for k in ('assign_ms','assign_frac_of_mfma_peak','iteration_ms')})
"; done
