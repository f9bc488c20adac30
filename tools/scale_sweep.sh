#!/bin/bash
# The multi-GPU measurements this repository could not take on its one-GPU development box: bench.py at N = 1, 2, 4, 8
# (the driver's contract line), then at N = max the two knobs of the K-split pipeline that only hardware can set --
# the column chunk of the reduce-scatters and how many channels (workgroups) RCCL may take from the GEMM.
# Usage: tools/scale_sweep.sh [max_gpus] [out_dir]      (one node; results as one JSON line per run)
set -u
MAXN=${1:-8}
OUT=${2:-gpurun_out/scale}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # run <n> <tag> [env assignments...]
  local n=$1 tag=$2; shift 2
  # the plain form: bench.py starts its own N ranks (torch.distributed.run ... bench.py --gpus N works as well)
  if [ "$n" = 1 ]; then
    env "$@" python "$ROOT/bench.py" --gpus 1 > "$OUT/n${n}_$tag.json" 2> "$OUT/n${n}_$tag.err"
  else
    env "$@" python "$ROOT/bench.py" --gpus "$n" --steps 10 --warmup 2 > "$OUT/n${n}_$tag.json" 2> "$OUT/n${n}_$tag.err"
  fi
  python - "$OUT/n${n}_$tag.json" "$n" "$tag" <<'PY'
import json, sys
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  b = d.get('dot_breakdown', {})
  print('N=%s %-22s %8.1f TFLOP/s  %8.2f ms/step  kernel-only %s  transport %s' % (
      sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], b.get('kernel_only_TFLOPs_whole_job'),
      d.get('comm', {}).get('transport')))
except Exception as e:
  print('N=%s %s: no result (%s) -- see the .err file' % (sys.argv[2], sys.argv[3], e))
PY
}
for n in 1 2 4 8; do
  [ "$n" -le "$MAXN" ] && run "$n" default
done
if [ "$MAXN" -gt 1 ]; then
  for cols in 2048 8192 16384; do run "$MAXN" "chunk$cols" SPARTAN_DOT_CHUNK_COLS=$cols; done
  for ch in 8 16 32; do run "$MAXN" "channels$ch" NCCL_MAX_NCHANNELS=$ch; done
fi
