"""The ONE line bench.py prints, kept small enough for the driver to parse.

bench.py collects a detailed record (every section, the launch table tools/roofline.py keys on, all emulation
cases).  That record goes to a side file and to stderr; what goes to stdout is `compact(record)`: the contract keys,
a compact `roofline`, `cpu_baseline`, and one short summary per extra section.  `fit(line)` is the last guard: a line
longer than LIMIT loses its optional sections, least important first, until it fits (`extras_truncated` names them).

Pure Python, no imports from the package: tests/test_bench_line.py builds worst-case records on a CPU box.
(The reference's runner this stands in for prints one line per benchmark too: tests/test_common.py:98-120.)
"""
import json

LIMIT = 7900             # bytes of the stdout line, hard bound: the driver did not parse r05's 21 KB line, and its
                         # kept stdout tail is 8018 characters -- the whole line fits inside it
TARGET = 6000            # what compact() is expected to stay under on a full default run (tests/test_bench_line.py)

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')
# optional sections in the order they are dropped when a line is too long (first = first to go)
DROP_ORDER = ('tile_store', 'launcher', 'comm', 'dot_f64', 'sparse', 'host', 'ksplit', 'collectives', 'tiles8',
              'hbm_dist', 'lreg_dist', 'kmeans_dist', 'rccl', 'dot_breakdown', 'lreg', 'kmeans', 'northstar', 'detail')


def _r(x, nd=3):
  return round(x, nd) if isinstance(x, float) else x


def _short(s, n):
  s = str(s)
  return s if len(s) <= n else s[:n - 3] + '...'


def _pick(d, keys):
  return {k: _r(d[k]) for k in keys if k in d and d[k] is not None}


def _roofline(full):
  rf = dict(full.get('roofline') or {})
  out = _pick(rf, ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'flop_per_launch',
                   'launches_per_step', 'traffic_floor', 'traffic_over_floor'))
  out.setdefault('traffic', None)
  if 'kernel' in rf:
    out['kernel'] = _short(rf['kernel'], 100)
  if rf.get('traffic') is None and rf.get('traffic_source'):
    out['traffic_note'] = _short(rf['traffic_source'], 160)
  ns = rf.get('northstar')
  if ns:
    out['northstar'] = _pick(ns, ('ms', 'TFLOPs', 'frac'))
    out['northstar']['shape'] = '32768^3 fp32, one tile'
  hs = rf.get('hbm_sections')
  if hs:
    out['hbm'] = {'copy_GBps': hs.get('measured_copy_GBps'), 'spec_GBps': hs.get('spec_GBps'),
                  'is': '{section: [GB/s, fraction of the copy rate measured in this run]}',
                  'sections': {_short(k, 40): [_r(v.get('GBps'), 1), v.get('frac_of_measured_copy')]
                               for k, v in (hs.get('sections') or {}).items()}}
  ka = rf.get('kmeans_assign')
  if ka:
    out['kmeans_assign'] = _pick(ka, ('bound', 'achieved', 'peak', 'unit', 'frac', 'ms', 'useful_fp32_TFLOPs'))
  for k in ('gemm_shapes',):
    if rf.get(k):
      out[k] = rf[k]
  return out


def _cpu(full):
  cb = full.get('cpu_baseline')
  if not cb:
    return None
  out = _pick(cb, ('value', 'unit', 'cores', 'kind', 'gemm_only_value', 'map_xx_plus_x_GBps', 'sum_axis0_GBps',
                   'wall_seconds'))
  out['sample'] = _short(cb.get('sample', ''), 520)
  for name in ('lreg', 'kmeans', 'lreg_step', 'kmeans_iteration'):
    sec = cb.get(name)
    if isinstance(sec, dict):
      out[name] = {k: (_short(v, 60) if isinstance(v, str) else _r(v, 4)) for k, v in sec.items()
                   if not isinstance(v, (dict, list))}
  if cb.get('scaled_for_memory'):
    out['scaled_for_memory'] = [_short(s, 120) for s in cb['scaled_for_memory'][:3]]
  return out


def _ksplit(sec):
  """Best case per p of the rank emulation: {p: [chunk_cols, step_ms, implied speed-up bound]}."""
  best = {}
  for c in sec.get('cases') or []:
    p = c.get('p')
    if p is None or 'implied_speedup_bound' not in c:
      continue
    if p not in best or c['implied_speedup_bound'] > best[p][2]:
      best[p] = [c.get('chunk_cols'), _r(c.get('step_ms')), c['implied_speedup_bound'],
                 c.get('gemm_slowdown_under_transfers')]
  out = {'best_per_p': {str(p): v for p, v in sorted(best.items())},
         'is': '{p: [chunk cols, step ms, implied speed-up bound, GEMM slowdown under transfers]}; upper bound, no links modelled'}
  out.update(_pick(sec, ('implied_8gpu_speedup_upper_bound', 'one_gpu_step_ms', 'error')))
  return out


def _flat(sec, keys=None, width=80):
  """A section as one level of scalars (strings shortened); nested dicts are kept only if small."""
  out = {}
  for k, v in sec.items():
    if keys is not None and k not in keys:
      continue
    if isinstance(v, str):
      out[k] = _short(v, width)
    elif isinstance(v, (dict, list)):
      if len(json.dumps(v)) <= 160:
        out[k] = v
    else:
      out[k] = _r(v, 4)
  return out


def compact(full):
  """The stdout line for a detailed record."""
  line = {}
  for k in CONTRACT:
    if k in full:
      line[k] = full[k]
  cfg = dict(full.get('config') or {})
  for k, v in list(cfg.items()):
    if isinstance(v, str):
      cfg[k] = _short(v, 260)
  line['config'] = cfg
  line['roofline'] = _roofline(full)
  cb = _cpu(full)
  if cb is not None:
    line['cpu_baseline'] = cb
  else:
    line.pop('cpu_baseline', None)
  for k in list(full):
    if k.startswith('northstar_') and isinstance(full[k], dict):
      line['northstar'] = _pick(full[k], ('workload', 'calls', 'ms_per_call', 'min_ms', 'TFLOPs', 'frac_of_mfma_peak', 'error'))
  if isinstance(full.get('lreg'), dict):
    line['lreg'] = _pick(full['lreg'], ('tile', 'steps', 'ms_per_step', 'step_kernels_ms', 'GBps', 'frac_of_measured_copy',
                                         'weights_finite', 'error'))
  if isinstance(full.get('kmeans'), dict):
    km = full['kmeans']
    line['kmeans'] = _pick(km, ('tile', 'assign_ms', 'assign_TFLOPs', 'assign_rechecked_points', 'accumulate_ms',
                                'accumulate_GBps', 'accumulate_launches', 'iteration_ms', 'fit_over_device',
                                'centers_finite', 'error'))
    split = km.get('assign_split') or {}
    line['kmeans'].update(_pick(split, ('frac_of_bf16_peak', 'frac_of_probed_bf16')))
  if isinstance(full.get('sparse'), dict):
    line['sparse'] = _pick(full['sparse'], ('tile', 'nnz', 'spmv_ms', 'spmv_GBps', 'spmv_frac_of_copy', 'five_iterations_ms',
                                             'pagerank_iteration_us', 'error'))
  if isinstance(full.get('host'), dict):
    line['host'] = _flat(full['host'])
  if isinstance(full.get('dot_f64'), dict):
    line['dot_f64'] = _pick(full['dot_f64'], ('ms_per_call', 'TFLOPs', 'peak_TFLOPs', 'frac_of_f64_mfma_peak', 'error'))
  if isinstance(full.get('ksplit_rank_emulation'), dict):
    line['ksplit'] = _ksplit(full['ksplit_rank_emulation'])
  t8 = full.get('one_gpu_8_tiles')
  if isinstance(t8, dict):
    line['tiles8'] = {'workers': t8.get('workers'), 'vs_one_tile': t8.get('vs_one_tile'),
                      'lreg_ms_per_step': (t8.get('lreg') or {}).get('ms_per_step'),
                      'kmeans_iteration_ms': (t8.get('kmeans') or {}).get('iteration_ms')}
    if 'error' in t8:
      line['tiles8']['error'] = _short(t8['error'], 160)
  if isinstance(full.get('tile_store'), dict):
    line['tile_store'] = full['tile_store']
  # several ranks
  if isinstance(full.get('dot_breakdown'), dict):
    line['dot_breakdown'] = {k: v for k, v in full['dot_breakdown'].items() if k != 'note'}
  if isinstance(full.get('collectives'), dict):
    line['collectives'] = full['collectives']
  if isinstance(full.get('rccl'), dict):
    line['rccl'] = _flat(full['rccl'], width=120)
    line['rccl'].pop('mapped', None)
    if isinstance(full['rccl'].get('mapped'), dict):       # how many copies of each runtime library the process maps
      line['rccl']['runtimes_mapped'] = {k: len(v) for k, v in full['rccl']['mapped'].items()}
  if isinstance(full.get('comm'), dict):
    line['comm'] = _flat(full['comm'], width=120)
  if 'launcher' in full:
    line['launcher'] = _short(full['launcher'], 80)
  if 'valid_scaling_measurement' in full:
    line['valid_scaling_measurement'] = full['valid_scaling_measurement']
  for name in ('hbm_dist', 'lreg_dist', 'kmeans_dist'):
    if isinstance(full.get(name), dict):
      line[name] = _flat(full[name])
  if full.get('detail'):
    line['detail'] = full['detail']
  return fit(line)


def fit(line, limit=LIMIT):
  """Drop optional sections (DROP_ORDER, then anything outside the contract, largest first) until the line's JSON
  is shorter than `limit`; the names dropped are listed under `extras_truncated`."""
  line = dict(line)
  dropped = []

  def size():
    if dropped:
      line['extras_truncated'] = dropped
    return len(json.dumps(line))
  if size() < limit:
    return line
  for name in DROP_ORDER:
    if name in line:
      del line[name]
      dropped.append(name)
      if size() < limit:
        return line
  extras = sorted((k for k in line if k not in CONTRACT and k != 'extras_truncated'),
                  key=lambda k: -len(json.dumps(line[k])))
  for name in extras:
    del line[name]
    dropped.append(name)
    if size() < limit:
      return line
  # only the contract is left: shorten inside roofline / cpu_baseline / config
  rf = line.get('roofline') or {}
  for name in ('hbm', 'gemm_shapes', 'kmeans_assign', 'northstar', 'traffic_note', 'kernel'):
    if name in rf:
      del rf[name]
      dropped.append('roofline.' + name)
      if size() < limit:
        return line
  cb = line.get('cpu_baseline') or {}
  for name in [k for k in cb if k not in ('value', 'unit', 'cores', 'kind', 'sample')]:
    del cb[name]
  if 'sample' in cb:
    cb['sample'] = _short(cb['sample'], 200)
  dropped.append('cpu_baseline.*')
  cfg = line.get('config') or {}
  for k in list(cfg):
    if k != 'workload':
      del cfg[k]
  if 'workload' in cfg:
    cfg['workload'] = _short(cfg['workload'], 200)
  dropped.append('config.*')
  size()
  return line
