# per-launch durations of the M-step update kernel over one timed cfg2 step (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
python - <<'PY'
import csv
rows = [r for r in csv.DictReader(open('/tmp/prof_t/t_kernel_trace.csv'))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = ('update_sums', 'assign_half_kernel', 'prep_fast32', 'm0_reduce', 'assign_split_rows', 'assign_requeue_rows', 'finalize_fx')
out = []
for r in rows:
  for n in names:
    if n in r['Kernel_Name']:
      out.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
half = len(out) // 2
for n, ms in out[half:]:
  print('%-22s %.3f ms' % (n, ms))
PY
