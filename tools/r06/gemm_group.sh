#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/gg
mkdir -p $OUT
for v in default g2 g4 g8; do
  LIB=$PWD/spartan_amd/csrc/libspartan_hip.so
  [ $v != default ] && LIB=$PWD/tools/r06/libspartan_hip_$v.so
  echo -n "$v: "; (cd tools; SPARTAN_HIP_LIB=$LIB python gemm_shapes.py 8192,8192,8192 32768,4096,4096 2>&1 | grep -v amdgpu | tr '\n' ' '); echo
  (cd /tmp; export TMPDIR=/tmp; SPARTAN_HIP_LIB=$LIB rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/$v --output-format csv -o p -- python $GRAFT_REPO_ROOT/bench.py --no-extras --steps 3 --warmup 1 > /dev/null 2> $OUT/$v.err)
  python3 - $OUT/$v <<'PY'
import sys,glob,csv,statistics
vals=[]
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'sp_gemm_glds_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': vals.append(float(r['Counter_Value']))
print('   FETCH_SIZE median %.0f KB over %d launches -> 2*FETCH = %.2f GB' % (statistics.median(vals), len(vals), 2*statistics.median(vals)*1024/1e9))
PY
done
rm -rf $OUT
