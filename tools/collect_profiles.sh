#!/bin/bash
# Copy what tools/profile_round.sh produced (gpurun_out/$ROUND, merged back from the GPU box) into profiles/.
ROUND=${1:-r03}
S=gpurun_out/$ROUND
cp $S/roofline.json profiles/${ROUND}_roofline.json
cp $S/bench_kernel_shapes.csv profiles/${ROUND}_bench_kernel_shapes.csv
cp $S/roofline_traffic.json profiles/roofline_traffic.json
cp $S/bench_n1.json profiles/${ROUND}_bench_n1.json
cp $S/bench_traced.json profiles/${ROUND}_bench_traced.json
cp $S/bench_detail.json profiles/${ROUND}_bench_detail.json
cp $S/bench_traced_detail.json profiles/${ROUND}_bench_traced_detail.json
cp $S/roofline.txt profiles/${ROUND}_roofline.txt
cp $S/trace/*kernel_trace.csv profiles/${ROUND}_bench_kernel_trace.csv
ls -la profiles/${ROUND}_* profiles/roofline_traffic.json
