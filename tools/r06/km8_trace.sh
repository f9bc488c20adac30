#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/km8
mkdir -p $OUT
cat > /tmp/km8.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np
import bench
import spartan_amd as sp
from spartan_amd import devarray as D
ctx = sp.initialize('hip', num_workers=8)
t0 = time.perf_counter()
out = bench.kmeans_dist_section(ctx, 8)
print(out)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o km -- python /tmp/km8.py > $OUT/out.txt 2> $OUT/err.txt
cd $GRAFT_REPO_ROOT
cat $OUT/out.txt | tail -2
f=$(ls $OUT/trace/*/*kernel_stats.csv $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import sys,csv
for x in list(csv.reader(open(sys.argv[1])))[1:34]:
    print("%-62s calls %5s total_us %9.0f avg_us %8.1f" % (x[0].replace("(anonymous namespace)::","").replace("void ","")[:62], x[1], float(x[2])/1e3, float(x[3])/1e3))
PY
rm -rf $OUT/trace
