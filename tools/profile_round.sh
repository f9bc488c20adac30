#!/bin/bash
# One GPU-box call: the driver's own bench command, plain and under rocprofv3 --kernel-trace (per-launch CSV), the
# counter passes (separate runs: --pmc only together with --kernel-trace) of the same command restricted to the
# sections whose kernels they are about, and tools/roofline.py over all of it.  Everything lands under
# gpurun_out/$ROUND/; copy roofline.json, bench_kernel_shapes.csv, roofline_traffic.json and bench_n1.json into
# profiles/ (roofline_traffic.json under exactly that name: bench.py quotes `traffic` from it when its tree hash
# matches the tree being benched).
set -u
ROUND=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SP_BENCH_DETAIL=$OUT/bench_detail.json python $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
wc -c $OUT/bench_n1.json
echo "bench rc=$?"
SP_BENCH_DETAIL=$OUT/bench_traced_detail.json rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o bench -- python $R/bench.py > $OUT/bench_traced.json 2> $OUT/bench_traced.err
echo "traced bench rc=$?"
for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_head_$tag --output-format csv -o p -- python $R/bench.py --no-extras --steps 3 --warmup 1 > /dev/null 2> $OUT/pmc_head_$tag.err
  rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_work_$tag --output-format csv -o p -- python $R/bench.py --steps 1 --warmup 0 --only hbm,lreg,kmeans,sparse > /dev/null 2> $OUT/pmc_work_$tag.err
done
cd $R
python tools/roofline.py --trace "$OUT/trace/**/*kernel_trace.csv" --bench $OUT/bench_traced_detail.json --pmc $OUT/pmc_head_* $OUT/pmc_work_* \
  --out-json $OUT/roofline.json --out-csv $OUT/bench_kernel_shapes.csv --traffic-json $OUT/roofline_traffic.json > $OUT/roofline.txt 2>&1
cat $OUT/roofline.txt
# keep the merged directory small: the raw per-launch trace of the whole bench is a few MB, the counter CSVs too
du -sh $OUT
