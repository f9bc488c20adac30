set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/spmv
mkdir -p $OUT
cd $R
for mode in "1 0 0" "1 0 1"; do
  set -- $mode
  SP_SPMV_PLANNED=$1 SP_SPMV_NT=$2 SP_SPMV_ABLATE=$3 python bench.py --steps 2 --warmup 1 --only sparse 2> $OUT/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planned=$1 nt=$2 ablate=$3', d['sparse']['spmv_ms'], d['sparse']['spmv_GBps'])"
done
