#!/bin/bash
# Per-kernel times of a LABELLED training-resolution call (48x256x28x28, 21 semantic x instance labels
# with an ignore value, as training passes them): rocprofv3 kernel stats.
cd /tmp && export TMPDIR=/tmp
cat > /tmp/small_lab.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from hsg_amd.utils.segsort import common as sc
x = torch.randn((48, 256, 28, 28), device='cuda:0')
sem = torch.randint(0, 21, (48, 28, 28), device='cuda:0')
inst = torch.randint(0, 6, (48, 28, 28), device='cuda:0')
lab = sem * 255 + inst
ign = int(lab.max()) + 1
lab = lab.masked_fill(sem == 20, ign)
for _ in range(12):
  out = sc.segment_by_kmeans(x, lab, [8, 8], ignore_index=ign, iterations=10)
torch.cuda.synchronize()
PY
rm -rf /tmp/prof_sl
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sl -o s -- python /tmp/small_lab.py > /dev/null 2>&1
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/prof_sl/s_kernel_stats.csv')))
tot = 0
for r in rows[:30]:
  per_call = float(r['TotalDurationNs']) / 12e3
  tot += per_call
  print('%-64s calls/call %5.1f avg %8.2f us  per call %8.2f us' % (r['Name'][:64], int(r['Calls']) / 12.0, float(r['AverageNs']) / 1e3, per_call))
print('sum per call %.1f us' % tot)
PY
