#!/bin/bash
# Counter passes over the interpreter tier's map kernel (5-op chain, 2 GiB tile): what the SIMDs do while the
# interpreted trip runs.  One GPU-box call; separate rocprofv3 runs per counter group (--pmc only with --kernel-trace).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-interp_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SP_NO_JIT=1 SP_NO_STATIC=1 INTERP_CASES=5op
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" \
           "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i --output-format csv -o p -- python $R/tools/interp_time.py > $OUT/g$i.log 2>&1
  echo "group $i rc=$?"
done
cd $R
python tools/pmc_summary.py $OUT DynProg > $OUT/summary.csv
cat $OUT/summary.csv
