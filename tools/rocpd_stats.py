"""Per-kernel statistics from a rocprofv3 results database (rocprofv3 --kernel-trace writes `<name>_results.db`
by default): name, calls, total / average / min / max duration in microseconds.  Usage:
  python tools/rocpd_stats.py gpurun_out/prof/x_results.db [substring ...] [--csv out.csv] [--trace]"""
import sqlite3
import sys


def main(argv):
  path = argv[1]
  pats = [a for a in argv[2:] if not a.startswith('--')]
  csv = argv[argv.index('--csv') + 1] if '--csv' in argv else None
  if csv in pats:
    pats.remove(csv)
  db = sqlite3.connect(path)
  cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
  name_col = 'name' if 'name' in cols else 'kernel_name'
  rows = db.execute('select %s, start, end from kernels order by start' % name_col).fetchall()
  if '--trace' in argv:
    t0 = rows[0][1] if rows else 0
    for name, s, e in rows:
      if not pats or any(p in name for p in pats):
        print('%12.1f us  +%10.1f us  %s' % ((s - t0) / 1e3, (e - s) / 1e3, name[:110]))
    return
  stats = {}
  for name, s, e in rows:
    st = stats.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    st[0] += 1
    st[1] += d
    st[2] = min(st[2], d)
    st[3] = max(st[3], d)
  out = ['"Name","Calls","TotalDurationUs","AverageUs","MinUs","MaxUs"']
  for name, (c, tot, mn, mx) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    if pats and not any(p in name for p in pats):
      continue
    out.append('"%s",%d,%.3f,%.3f,%.3f,%.3f' % (name, c, tot / 1e3, tot / c / 1e3, mn / 1e3, mx / 1e3))
  text = '\n'.join(out)
  if csv:
    open(csv, 'w').write(text + '\n')
  print(text)


if __name__ == '__main__':
  main(sys.argv)
