"""Roofline fractions recomputed from a rocprofv3 kernel trace of `python bench.py`.

  rocprofv3 --kernel-trace --output-format csv -d DIR -o bench -- python bench.py > bench.json
  python tools/roofline.py --trace DIR/**/bench_kernel_trace.csv --bench bench.json \
         [--pmc PMC_DIR ...] --out-json profiles/rNN_roofline.json --out-csv profiles/rNN_bench_kernel_shapes.csv

Two outputs, both derived from the per-launch rows of the trace (one row per dispatch: name, grid, start, end):

  * the CSV: Calls / Total / Average / Min / Max duration per (kernel, grid size, workgroup size) -- launches of one
    kernel on different problem sizes are separate rows (the GEMM at 8192^3 and at 32768^3 have different grids);
  * the JSON: for every timed section of the bench line's `profile_table` (label, kernel, launches, algorithmic
    units per launch, host-clock window) the launches of that kernel that started inside the window, their count
    (must equal `launches`), the AVERAGE duration, and achieved = units / average against the peak that bounds the
    kernel (MFMA fp32 157.3 TFLOP/s; HBM 8000 GB/s spec, and the copy rate measured in the same run).  With --pmc,
    the counter passes of tools/profile_round.sh add HBM traffic per launch = 2 * FETCH_SIZE + WRITE_SIZE (KB;
    the gfx950 correction calibrated in profiles/pmc_traffic.json: FETCH_SIZE reports half the bytes of wide
    coalesced reads) and MFMA busy, stamped with the kernel sources' hash, and profiles/roofline_traffic.json is
    rewritten for bench.py to quote.

The windows are host times in several clocks (bench.py records CLOCK_MONOTONIC, BOOTTIME, MONOTONIC_RAW and
wall time); which one rocprofv3's timestamps use on the box is found by counting: the clock under which the windows
contain the expected numbers of launches.
"""
import argparse
import csv
import glob
import json
import os
import re
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBPS = 8000.0


def short_name(name):
  name = name.replace('(anonymous namespace)::', '')
  name = re.sub(r'^void ', '', name)
  depth, out = 0, ''
  for ch in name:            # cut the argument list: the first '(' outside template brackets
    if ch == '<':
      depth += 1
    elif ch == '>':
      depth -= 1
    elif ch == '(' and depth == 0:
      break
    out += ch
  return out.strip()


def load_trace(paths):
  rows = []
  for path in paths:
    for r in csv.DictReader(open(path)):
      rows.append({'name': short_name(r['Kernel_Name']), 'start': int(r['Start_Timestamp']), 'end': int(r['End_Timestamp']),
                   'grid': int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1),
                   'wg': int(r['Workgroup_Size_X']), 'vgpr': r.get('VGPR_Count', ''), 'lds': r.get('LDS_Block_Size', '')})
  rows.sort(key=lambda r: r['start'])
  return rows


def shape_stats(rows):
  groups = {}
  for r in rows:
    groups.setdefault((r['name'], r['grid'], r['wg']), []).append((r['end'] - r['start']) / 1e3)
  out = []
  for (name, grid, wg), d in groups.items():
    out.append({'name': name, 'grid_threads': grid, 'workgroup': wg, 'calls': len(d), 'total_us': sum(d),
                'avg_us': sum(d) / len(d), 'min_us': min(d), 'max_us': max(d)})
  out.sort(key=lambda g: -g['total_us'])
  return out


def pick_clock(rows, table):
  best = None
  for clock in ('monotonic', 'boottime', 'monotonic_raw', 'realtime'):
    good = 0
    for e in table:
      if clock not in e.get('t0', {}):
        continue
      n = sum(1 for r in rows if e['kernel'] in r['name'] and e['t0'][clock] <= r['start'] <= e['t1'][clock])
      good += (n == e['launches'])
    if best is None or good > best[1]:
      best = (clock, good)
  return best


def pmc_values(dirs):
  """{short kernel name: {counter: median value over launches (first launch of each kernel dropped when there are more)}}"""
  vals = {}
  for d in dirs:
    if os.path.isfile(d) and not d.endswith('.csv'):
      continue
    files = [d] if os.path.isfile(d) else glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    for f in files:
      for r in csv.DictReader(open(f)):
        key = (short_name(r['Kernel_Name']), int(r['Grid_Size']) if r.get('Grid_Size') else 0)
        vals.setdefault(key, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
  out = {}
  for key, ctrs in vals.items():
    out[key] = {c: statistics.median(v[1:] or v) for c, v in ctrs.items()}
    out[key]['_launches'] = max(len(v) for v in ctrs.values())
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--trace', nargs='+', required=True)
  ap.add_argument('--bench', required=True)
  ap.add_argument('--pmc', nargs='*', default=[])
  ap.add_argument('--out-json')
  ap.add_argument('--out-csv')
  ap.add_argument('--traffic-json', help='rewrite this file (profiles/roofline_traffic.json) from the --pmc passes')
  args = ap.parse_args()
  paths = [p for pat in args.trace for p in glob.glob(pat, recursive=True)]
  rows = load_trace(paths)
  line = json.loads([l for l in open(args.bench).read().splitlines() if l.startswith('{')][-1])
  table = line.get('profile_table', [])
  shapes = shape_stats(rows)
  if args.out_csv:
    with open(args.out_csv, 'w') as f:
      f.write('"Name","GridThreads","WorkgroupSize","Calls","TotalDurationUs","AverageUs","MinUs","MaxUs"\n')
      for g in shapes:
        if g['name'].startswith('sp_') or 'sp_' in g['name']:
          f.write('"%s",%d,%d,%d,%.3f,%.3f,%.3f,%.3f\n' % (g['name'], g['grid_threads'], g['workgroup'], g['calls'],
                                                          g['total_us'], g['avg_us'], g['min_us'], g['max_us']))
  clock, good = pick_clock(rows, table) if table else (None, 0)
  copy_gbps = line.get('hbm', {}).get('stream_copy_GBps')
  sections = []
  for e in table:
    sel = [r for r in rows if e['kernel'] in r['name'] and clock and e['t0'][clock] <= r['start'] <= e['t1'][clock]]
    d = [(r['end'] - r['start']) / 1e3 for r in sel]
    rec = {'label': e['label'], 'kernel': sorted(set(r['name'] for r in sel)), 'grid_threads': sorted(set(r['grid'] for r in sel)),
           'launches_expected': e['launches'], 'launches_in_trace': len(sel), 'units_per_launch': e['units_per_launch'],
           'unit': e['unit'], 'bound': e['bound']}
    if d:
      avg = sum(d) / len(d)
      rec.update({'avg_us': round(avg, 3), 'min_us': round(min(d), 3), 'max_us': round(max(d), 3)})
      if e['unit'] == 'flop':
        ach = e['units_per_launch'] / (avg * 1e-6) / 1e12
        peak = {'mfma_bf16': MFMA_BF16_PEAK_TFLOPS, 'mfma_f64': line.get('dot_f64', {}).get('peak_TFLOPs', 78.6)}.get(e['bound'], MFMA_F32_PEAK_TFLOPS)
        rec.update({'achieved': round(ach, 2), 'achieved_unit': 'TFLOP/s', 'peak': peak, 'frac': round(ach / peak, 4)})
      else:
        ach = e['units_per_launch'] / (avg * 1e-6) / 1e9
        rec.update({'achieved': round(ach, 1), 'achieved_unit': 'GB/s', 'peak': HBM_PEAK_GBPS,
                    'frac': round(ach / HBM_PEAK_GBPS, 4)})
        if copy_gbps:
          rec['frac_of_measured_copy'] = round(ach / copy_gbps, 3)
    sections.append(rec)
  from spartan_amd import _hip
  out = {'tree_sha': _hip.source_sha(), 'trace_rows': len(rows), 'timestamp_clock': clock,
         'sections_with_expected_launch_count': '%d of %d' % (good, len(table)), 'bench_value': line.get('value'),
         'bench_roofline': line.get('roofline'), 'sections': sections}
  if args.pmc:
    pm = pmc_values(args.pmc)
    traffic = {'tree_sha': out['tree_sha'],
               '_note': 'rocprofv3 --kernel-trace --pmc passes of tools/profile_round.sh (separate runs per counter group); '
                        'traffic_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction calibrated in '
                        'profiles/pmc_traffic.json'}
    counters = []
    for (name, grid), c in sorted(pm.items(), key=lambda kv: str(kv[0])):
      if 'sp_' not in name:
        continue
      rec = {'kernel': name, 'grid_threads': grid, 'launches': c['_launches']}
      if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
        rec.update({'FETCH_SIZE_KB': c['FETCH_SIZE'], 'WRITE_SIZE_KB': c['WRITE_SIZE'],
                    'traffic_bytes': int((2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024)})
      if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c and c['GRBM_GUI_ACTIVE']:
        rec['mfma_busy_fraction_of_1024_simds'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['GRBM_GUI_ACTIVE'] / 8.0), 4)
        rec['SQ_VALU_MFMA_BUSY_CYCLES'] = c['SQ_VALU_MFMA_BUSY_CYCLES']
        rec['GRBM_GUI_ACTIVE_sum_of_8_xcds'] = c['GRBM_GUI_ACTIVE']
      counters.append(rec)
      if 'sp_gemm_glds_kernel' in name and 'traffic_bytes' in rec:
        tiles = grid // 256
        for n in (8192, 32768):
          if tiles == (n // 256) * (n // 128):
            traffic['gemm_%d' % n] = {'traffic_bytes': rec['traffic_bytes'], 'algorithmic_min_bytes': 12 * n * n,
                                      'kernel': name, 'FETCH_SIZE_KB': c['FETCH_SIZE'], 'WRITE_SIZE_KB': c['WRITE_SIZE'],
                                      'mfma_busy': rec.get('mfma_busy_fraction_of_1024_simds')}
    out['counters'] = counters
    if args.traffic_json:
      json.dump(traffic, open(args.traffic_json, 'w'), indent=1)
  text = json.dumps(out, indent=1)
  if args.out_json:
    open(args.out_json, 'w').write(text + '\n')
  for s in sections:
    print('%-34s launches %3d/%-3d avg %10.2f us  %s' % (s['label'], s['launches_in_trace'], s['launches_expected'], s.get('avg_us', 0),
                                                         ('%.2f %s = %.3f of peak' % (s['achieved'], s['achieved_unit'], s['frac'])) if 'frac' in s else ''))
  print('clock: %s (%s sections matched)' % (clock, out['sections_with_expected_launch_count']))


if __name__ == '__main__':
  main()
