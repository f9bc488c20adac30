"""Target of the `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes for the reduce, arg-reduce and k-means
kernels: each program of bench.py's hbm / kmeans sections is run three times on the same tiles.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_red_f --output-format csv -- python tools/reduce_pmc.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_red_w --output-format csv -- python tools/reduce_pmc.py
  python tools/reduce_pmc.py --summarise gpurun_out/pmc_red_f gpurun_out/pmc_red_w   # -> per-kernel traffic (JSON)

Traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB; gfx950 correction calibrated in profiles/pmc_traffic.json)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROWS, COLS = 8192, 65536
KM = (1250000, 1024, 256)
# kernel-name fragment -> algorithmic bytes per launch
ALGO = {
    'sp_reduce_cols_kernel': 4 * ROWS * COLS,
    'sp_reduce_rows': 4 * ROWS * COLS,
    'sp_arg': 4 * ROWS * COLS,
    'sp_nearest_fused': 4 * KM[0] * KM[2],
    'sp_segment_sum_kernel': 4 * KM[0] * KM[2],
}


def run():
  import numpy as np
  import torch
  import spartan_amd as sp
  from spartan_amd import kernels
  from bench import SEED, device_uniform
  ctx = sp.initialize('hip', num_workers=1)
  X = sp.from_tile_fn((ROWS, COLS), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 7)).force()
  Xv = sp.Val(val=X)
  for _ in range(3):
    for axis in (None, 0, 1):
      sp.sum(Xv, axis).force()
    sp.argmax(Xv, 1).force()
  torch.cuda.synchronize()
  del X, Xv
  torch.cuda.empty_cache()
  n, k, d = KM
  P = sp.from_tile_fn((n, d), np.float32, lambda ex: device_uniform(ex, 0.0, 1.0, SEED + 21)).force()
  x = ctx.tile(list(P.tiles.values())[0]).data
  cdev = ctx.backend.from_numpy(np.random.RandomState(SEED).rand(k, d))
  labels = torch.empty(n, dtype=torch.int64, device=x.device)
  sums = torch.empty(k, d, dtype=torch.float32, device=x.device)
  counts = torch.empty(k, dtype=torch.int64, device=x.device)
  for _ in range(3):
    kernels.nearest_center(x, cdev, labels)
    kernels.bincount(labels, k, counts)
    kernels.segment_sum(x, labels, k, sums)
  torch.cuda.synchronize()
  print('done')


def summarise(fetch_dir, write_dir):
  def load(d, name):
    per = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
      for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != name:
          continue
        kn = r['Kernel_Name']
        if 'sp_' not in kn:
          continue
        short = kn.split('sp_', 1)[1].split('(')[0]
        per.setdefault('sp_' + short, []).append(float(r['Counter_Value']))
    return per
  F, W = load(fetch_dir, 'FETCH_SIZE'), load(write_dir, 'WRITE_SIZE')
  out = {}
  for kn in sorted(F):
    f, w = F[kn], W.get(kn, [0.0])
    # skip the first launch of each kernel (cold caches, first-touch pages): report the median of the rest
    f2, w2 = sorted(f[1:] or f), sorted(w[1:] or w)
    fk, wk = f2[len(f2) // 2], w2[len(w2) // 2]
    out[kn] = {'launches': len(f), 'FETCH_SIZE_KB': fk, 'WRITE_SIZE_KB': wk,
               'traffic_bytes': int((2 * fk + wk) * 1024)}
    for frag, b in ALGO.items():
      if frag in kn and b:
        out[kn]['algorithmic_bytes'] = b
        out[kn]['traffic_over_algorithmic'] = round(out[kn]['traffic_bytes'] / b, 3)
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == '--summarise':
    summarise(sys.argv[2], sys.argv[3])
  else:
    run()
