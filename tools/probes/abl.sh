cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abl -o a -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --reps 3 --only e --lloyd-warm 0 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/abl/a_kernel_stats.csv")):
    if "assign_split" in r["Name"]:
        print("split kernel avg ms", round(float(r["AverageNs"])/1e6,4))
PY
