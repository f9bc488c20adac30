#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/kmfit
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o km -- python $GRAFT_REPO_ROOT/bench.py --only kmeans --steps 2 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
f=$(ls $OUT/trace/*/*kernel_trace.csv $OUT/trace/*kernel_trace.csv 2>/dev/null | head -1)
python tools/gaps.py $f 3 > $OUT/gaps.txt
python - "$f" <<'PY' > $OUT/iter.txt
import csv,sys
rows=sorted(({'n':r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:70],'s':int(r['Start_Timestamp']),'e':int(r['End_Timestamp'])} for r in csv.DictReader(open(sys.argv[1]))),key=lambda r:r['s'])
# last fit: find last 2 occurrences of the first-pass kernel
idx=[i for i,r in enumerate(rows) if 'sp_nearest_split_kernel<false, false' in r['n']]
a,b=idx[-2],idx[-1]
# walk back from a to the start of its iteration: first kernel after previous segment_combine
start=a
while start>0 and 'segment_combine' not in rows[start-1]['n']: start-=1
end=b
while end>0 and 'segment_combine' not in rows[end-1]['n']: end-=1
prev=None
for r in rows[start:end]:
    gap=(r['s']-prev['e'])/1e3 if prev else 0
    print('%8.1f us gap %6.1f  %s'%((r['e']-r['s'])/1e3,gap,r['n']))
    prev=r
print('iteration span %.1f us'%((rows[end-1]['e']-rows[start]['s'])/1e3))
PY
tail -3 $OUT/gaps.txt; cat $OUT/iter.txt
rm -rf $OUT/trace
