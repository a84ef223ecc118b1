# per-kernel stats of one bench.py run (3 steps): rocprofv3 --kernel-trace --stats, CSV summary
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-exchange --cpu-images 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/prof/b_kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.4f ms total %9.2f ms %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
