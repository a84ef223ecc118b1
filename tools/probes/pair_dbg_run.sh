cd /tmp; export TMPDIR=/tmp
for v in 1 2; do
  cp $GRAFT_REPO_ROOT/build_dbg/libhsgk_dbg$v.so $GRAFT_REPO_ROOT/hsg_amd/csrc/libhsgk.so
  rm -rf /tmp/px
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -o x -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --steps 3 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
  python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/px/x_kernel_stats.csv")))
for r in rows[:3]:
  if "pair" in r["Name"]: print("debug $v", r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"])
PY
done
