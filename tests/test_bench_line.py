"""bench.py's stdout line stays small enough for the driver to parse (a 21 KB line was not parsed).

The records here are built on the CPU: the round-5 record kept under profiles/ (the one that was too long), and a
synthetic worst case in which every section carries far more than any run produces."""
import json
import os

import pytest

from tools import bench_line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _contract_ok(line):
  for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
    assert k in line, k
  rf = line['roofline']
  for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
    assert k in rf, k
  assert 'workload' in line['config']


def test_round5_record_compacts_under_the_target():
  full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_n1.json')))
  assert len(json.dumps(full)) > 20000          # the line the driver could not parse
  line = bench_line.compact(full)
  text = json.dumps(line)
  assert len(text) < bench_line.TARGET, len(text)
  assert 'extras_truncated' not in line
  _contract_ok(line)
  # the numbers the judge reads survive the compaction
  assert line['value'] == full['value'] and line['roofline']['frac'] == pytest.approx(full['roofline']['frac'], abs=1e-3)
  assert line['cpu_baseline']['cores'] == full['cpu_baseline']['cores']
  assert line['cpu_baseline']['kind'] == 'port' and line['cpu_baseline']['sample']
  secs = line['roofline']['hbm']['sections']
  assert secs['sum_axis0'] == [full['hbm']['sum_axis0_GBps'], full['hbm']['frac_of_measured_copy']['sum_axis0_GBps']]
  assert line['roofline']['northstar']['TFLOPs'] == full['northstar_32768']['TFLOPs']
  assert line['kmeans']['assign_ms'] == full['kmeans']['assign_ms']
  assert set(line['ksplit']['best_per_p']) == {'2', '4', '8'}
  assert 'profile_table' not in line and 'hbm' not in line


def _worst_case():
  blob = 'x' * 4000
  sections = {('section_%03d_with_a_long_name' % i): {'GBps': 6543.21, 'frac_of_measured_copy': 0.987, 'frac_of_spec': 0.8}
              for i in range(80)}
  full = {
      'metric': 'spartan.dot TFLOP/s (+ map/reduce HBM GB/s)', 'value': 150.0, 'unit': 'TFLOP/s', 'n_gpus': 8, 'steps': 20,
      'warmup': 3, 'ms_per_step': 7.3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
      'data': 'synthetic',
      'config': {'workload': blob, 'parallelism': blob, 'inputs': blob, 'flop_per_step': 1.0},
      'roofline': {'bound': 'mfma', 'kernel': blob, 'achieved': 150.0, 'peak': 157.3, 'unit': 'TFLOP/s', 'frac': 0.95,
                   'traffic': None, 'traffic_source': blob, 'northstar': {'ms': 470.0, 'TFLOPs': 149.0, 'frac': 0.95},
                   'hbm_sections': {'measured_copy_GBps': 6500.0, 'spec_GBps': 8000.0, 'sections': sections},
                   'kmeans_assign': {'bound': 'mfma_bf16', 'achieved': 1.0, 'peak': 2500.0, 'frac': 0.4, 'frac_is': blob},
                   'gemm_shapes': {('%dx%dx%d' % (i, i, i)): [100.0, 0.9] for i in range(1000, 1400)}},
      'cpu_baseline': {'value': 0.5, 'unit': 'TFLOP/s', 'cores': 64, 'kind': 'port', 'sample': blob, 'value_is': blob,
                       'lreg': {'shape': blob, 'seconds': 1.0}, 'kmeans': {'shape': blob, 'seconds': 1.0},
                       'scaled_for_memory': [blob] * 10},
      'profile_table': [{'label': blob, 't0': {'a': 1}, 't1': {'a': 2}}] * 60,
      'hbm': {('k%d_GBps' % i): 1.0 for i in range(300)},
      'ksplit_rank_emulation': {'cases': [{'p': p, 'chunk_cols': c, 'step_ms': 1.0, 'implied_speedup_bound': 1.0 + c * 1e-6,
                                           'gemm_slowdown_under_transfers': 1.01, 'junk': blob}
                                          for p in (2, 4, 8) for c in (1024, 2048, 4096, 8192)], 'note': blob},
      'one_gpu_8_tiles': {'workers': 8, 'hbm': {'junk': blob}, 'vs_one_tile': {('k%d' % i): 1.0 for i in range(200)}},
      'lreg': {'tile': blob, 'note': blob, 'ms_per_step': 0.3}, 'kmeans': {'tile': blob, 'assign_ms': 1.0},
      'sparse': {'tile': blob}, 'host': {('k%d' % i): blob for i in range(30)},
      'dot_breakdown': {('k%d' % i): 1.0 for i in range(100)}, 'collectives': {('k%d' % i): [1.0, 2.0, 3.0] for i in range(300)},
      'rccl': {('k%d' % i): blob for i in range(30)}, 'comm': {('k%d' % i): blob for i in range(30)},
      'hbm_dist': {('k%d' % i): blob for i in range(30)}, 'lreg_dist': {'a': blob}, 'kmeans_dist': {'a': blob},
      'launcher': blob, 'detail': 'gpurun_out/bench_detail_n8.json', 'northstar_32768': {'workload': blob, 'TFLOPs': 1.0},
      'something_new': {'a': blob},
  }
  return full


def test_worst_case_record_is_bounded_and_keeps_the_contract():
  line = bench_line.compact(_worst_case())
  text = json.dumps(line)
  assert len(text) < bench_line.LIMIT, len(text)
  _contract_ok(line)
  assert line['extras_truncated']                 # it had to drop sections, and says which
  assert line['cpu_baseline']['cores'] == 64 and line['cpu_baseline']['kind'] == 'port'
  assert line['value'] == 150.0 and line['roofline']['frac'] == 0.95


def test_fit_drops_least_important_first_and_unknown_keys_too():
  line = {'metric': 'm', 'value': 1.0, 'roofline': {'bound': 'mfma'}, 'tile_store': {'a': 'y' * 3000},
          'kmeans': {'a': 'y' * 3000}, 'unknown': {'a': 'y' * 6000}}
  out = bench_line.fit(line, limit=8000)
  assert 'tile_store' not in out and 'kmeans' not in out and 'unknown' in out
  assert out['extras_truncated'] == ['tile_store', 'kmeans']
  out = bench_line.fit(line, limit=3000)
  assert set(out) == {'metric', 'value', 'roofline', 'extras_truncated'}


def test_emit_writes_the_compact_line_and_the_detail_file(tmp_path, monkeypatch, capfd):
  import bench
  full = json.load(open(os.path.join(ROOT, 'profiles', 'r05_bench_n1.json')))
  detail = tmp_path / 'detail.json'
  monkeypatch.setenv('SP_BENCH_DETAIL', str(detail))
  monkeypatch.setattr(bench, '_REAL_STDOUT', None)
  bench._emit(full, 0)
  out, err = capfd.readouterr()
  lines = [l for l in out.splitlines() if l.startswith('{')]
  assert len(lines) == 1 and len(lines[0]) < bench_line.LIMIT
  line = json.loads(lines[0])
  _contract_ok(line)
  assert line['detail'] == str(detail)
  kept = json.load(open(str(detail)))
  assert 'profile_table' in kept and kept['hbm'] == full['hbm']
  assert len(kept['ksplit_rank_emulation']['cases']) == len(full['ksplit_rank_emulation']['cases'])
  bench._emit(full, 1)                           # other ranks print nothing
  out, err = capfd.readouterr()
  assert out == ''
