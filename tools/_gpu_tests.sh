cd /root/repo/tools
for s in 2 3 4 5; do timeout 900 python fuzz_kmeans.py $s 2>&1 | tail -3; done
cd /root/repo
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
    h=find(d); print(json.dumps({k:v for k,v in h.items() if 'assign' in k or 'iteration_ms' in k}))
"
