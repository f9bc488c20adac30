#!/bin/bash
# One GPU-box call: the bench line, its rocprofv3 kernel statistics, and the PMC passes (separate runs, counters
# only with --kernel-trace) of the two MFMA kernels.  Everything lands under gpurun_out/r02/; tools/rocpd_stats.py
# and tools/pmc_summary.py turn it into the small files kept under profiles/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
rocprofv3 --kernel-trace --stats -d $OUT/bench_prof -o bench -- python $R/bench.py > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
for kind in gemm km; do
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $ctr | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $ctr -d $OUT/pmc_${kind}_$tag --output-format csv -o $kind -- python $R/tools/${kind}_pmc.py > /dev/null 2> $OUT/pmc_${kind}_$tag.err
  done
done
ls -R $OUT | head -60
