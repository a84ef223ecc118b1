# final evidence of the round at the last source commit: GPU suite, smoke, the bench lines of every workload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles; mkdir -p $O; cd $R
commit=$(cat .profile_commit 2>/dev/null || echo unknown)
{ echo "# python -m pytest tests -q -m gpu at $commit (library rebuilt from sources on the box: make -C hsg_amd/csrc clean all torch)"
  make -s -C hsg_amd/csrc clean > /dev/null 2>&1; make -s -j32 -C hsg_amd/csrc all torch 2>&1 | tail -2
  timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
  echo "# python -c 'import __graft_entry__ as g; g.smoke()'"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2; } > $O/r06_final_gputest.txt 2>&1
cat $O/r06_final_gputest.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/r06_final_bench.json
for wl in cfg3 cfg4 cfg5; do timeout 400 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-images 0 2>/dev/null | tail -1 > $O/r06_final_bench_$wl.json; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/profiles/r06_final_bench*.json')):
  d = json.loads(open(f).read())
  print(f.split('/')[-1], d['ms_per_step'], round(d['value'] / 1e6, 1), 'Mpx/s frac', d['roofline']['frac'], d['roofline']['bytes_source'][:34], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
{ for p in train_step_wall train_step_gpu train_step_syncs train_step_gaps train_step_ctypes; do echo "== tools/probes/$p.py"; timeout 300 python -u tools/probes/$p.py 2>&1 | grep -v -i "amdgpu.ids\|warn"; done; } > $O/r06_train_step.txt
timeout 300 python tools/probes/train_step_lines.py 2>&1 | grep -v -i "warn\|amdgpu" > $O/r06_train_step_lines.txt
head -3 $O/r06_train_step.txt
