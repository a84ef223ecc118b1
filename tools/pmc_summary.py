#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch of
every counter for kernels whose name contains a filter string."""
import collections
import csv
import sys

path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'hsgk')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
  if filt in r['Kernel_Name']:
    agg[r['Kernel_Name'].split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
  print(k)
  for c, xs in sorted(v.items()):
    print('   %-28s %16.0f  (n=%d)' % (c, sum(xs) / len(xs), len(xs)))
