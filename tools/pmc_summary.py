#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch of
every counter for kernels whose name contains a filter string."""
import collections
import csv
import re
import sys

path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'hsgk')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
  if filt in r['Kernel_Name']:
    m = re.search(r'([a-z0-9_]+_kernel)', r['Kernel_Name'])
    agg[m.group(1) if m else r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
skip = set(sys.argv[3].split(',')) if len(sys.argv) > 3 else set()
for k, v in agg.items():
  if k in skip:
    continue
  print(k)
  for c, xs in sorted(v.items()):
    print('   %-28s %16.0f  (n=%d)' % (c, sum(xs) / len(xs), len(xs)))
