cd /root/repo/tools
timeout 900 python fuzz_kmeans.py 41 2>&1 | tail -2
cd /root/repo
timeout 100 python tools/km_first.py 2>&1 | tail -1
timeout 300 python bench.py --only kmeans 2>/dev/null | python -c "
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
    h=find(d); print(json.dumps({k:v for k,v in h.items() if k in ('assign_ms','assign_standalone_ms','assign_rechecked_points','iteration_ms','assign_fp32_tier')}))
"
