# per-iteration durations of the E-step kernels of one bench workload (kernel trace of the LAST call):
#   bash tools/probes/cfg_iter_trace.sh cfg4 [iid|mixture]
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_it
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_it -o t -- python $GRAFT_REPO_ROOT/bench.py --workload ${1:-cfg4} --flavour ${2:-iid} --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_it/**/t_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = ('assign_half', 'assign_requeue', 'assign_split', 'assign_hard', 'update_sums', 'prep_fast', 'm0_reduce')
sel = [r for r in rows if any(n in r['Kernel_Name'] for n in names)]
# the last call = after the last prep kernel
last = max(i for i, r in enumerate(sel) if 'prep_fast' in r['Kernel_Name'])
for r in sel[last:]:
  nm = r['Kernel_Name']
  nm = nm[nm.find('hsgk') + 4:][:40]
  print('%-42s %8.1f us' % (nm, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
