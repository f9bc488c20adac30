"""Idle gaps of the GPU in a rocprofv3 kernel trace: where the device waited for the host.
python tools/gaps.py TRACE.csv [min_gap_us] [first_kernel_substring]
Prints every gap longer than min_gap_us between the end of one kernel and the start of the next (with the two kernel
names), and the busy / idle split of the traced interval (from the first launch matching the substring, if given)."""
import csv
import sys

rows = sorted(({'name': r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60], 's': int(r['Start_Timestamp']), 'e': int(r['End_Timestamp'])}
               for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r['s'])
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
if len(sys.argv) > 3:
  first = next(i for i, r in enumerate(rows) if sys.argv[3] in r['name'])
  rows = rows[first:]
busy = idle = 0
end = rows[0]['s']
hist = {}
for prev, r in zip([None] + rows[:-1], rows):
  gap = (r['s'] - end) / 1e3
  if prev is not None and gap > 0:
    idle += gap
    if gap >= min_gap:
      key = (prev['name'], r['name'])
      h = hist.setdefault(key, [0, 0.0])
      h[0] += 1
      h[1] += gap
  busy += max(0, r['e'] - max(end, r['s'])) / 1e3
  end = max(end, r['e'])
for (a, b), (cnt, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
  print('%4d x %9.1f us avg  after %-50s before %s' % (cnt, tot / cnt, a, b))
print('busy %.3f ms, idle %.3f ms (%.1f %%) over %d launches' % (busy / 1e3, idle / 1e3, 100 * idle / (busy + idle), len(rows)))
