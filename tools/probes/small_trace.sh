#!/bin/bash
# Kernel timeline of one training-resolution call (48x256x28x28, K = 8x8, 10 iterations):
# per-kernel stats and the start-to-end span of one call from the rocprofv3 kernel trace.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/probes/small_one.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/prof_s/s_kernel_stats.csv')))
for r in rows[:16]:
  print('%-60s calls %5s avg %8.2f us total %8.2f ms' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
tr = list(csv.DictReader(open('/tmp/prof_s/s_kernel_trace.csv')))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
# the last call: kernels after the last prep kernel
idx = [i for i, r in enumerate(tr) if 'prep_' in r['Kernel_Name']]
last = idx[-1]
first = last
while first > 0 and 'prep_' not in tr[first - 1]['Kernel_Name'] and (int(tr[first]['Start_Timestamp']) - int(tr[first - 1]['End_Timestamp'])) < 200000:
  first -= 1
seq = tr[first:]
t0 = int(seq[0]['Start_Timestamp'])
busy = 0
for r in seq:
  s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
  busy += e - s
print('last call: %d kernels, span %.1f us, sum of kernel durations %.1f us' % (len(seq), (int(seq[-1]['End_Timestamp']) - t0) / 1e3, busy / 1e3))
for r in seq[:40]:
  s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
  print('  %8.1f -> %8.1f us (%6.1f)  %s' % (s / 1e3, e / 1e3, (e - s) / 1e3, r['Kernel_Name'][:70]))
PY
