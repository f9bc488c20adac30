"""Per-kernel medians of the counters in rocprofv3 `*_counter_collection.csv` files (--pmc passes).
Usage: python tools/pmc_summary.py <dir or csv> [kernel-name substring ...]   -> CSV on stdout"""
import csv
import glob
import os
import statistics
import sys


def main(argv):
  src = argv[1]
  pats = argv[2:]
  files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
  vals = {}
  for f in files:
    for row in csv.DictReader(open(f)):
      name = row['Kernel_Name']
      if pats and not any(p in name for p in pats):
        continue
      short = name.replace('(anonymous namespace)::', '').split('(')[0][-90:]
      key = (short, row['Counter_Name'])
      vals.setdefault(key, []).append((float(row['Counter_Value']), int(row['End_Timestamp']) - int(row['Start_Timestamp']),
                                       row['VGPR_Count'], row['LDS_Block_Size']))
  print('"Kernel","Counter","Launches","MedianValue","MedianDurationNs","VGPRs","LDS_Block_Size"')
  for (short, ctr), v in sorted(vals.items()):
    print('"%s","%s",%d,%.1f,%d,%s,%s' % (short, ctr, len(v), statistics.median(x[0] for x in v),
                                          statistics.median(x[1] for x in v), v[0][2], v[0][3]))


if __name__ == '__main__':
  main(sys.argv)
