# ablation of the fp16 filter kernel: rebuild the library ON the GPU box with -DHSGK_ABL=n
for a in 5 6 7; do
  touch hsg_amd/csrc/kmeans.hip; make -C hsg_amd/csrc EXTRA=-DHSGK_ABL=$a -j8 > /dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ablh -o a$a -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --reps 5 --only e --unit 2 > /dev/null 2>&1)
  python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ablh/a$a" + "_kernel_stats.csv")):
    if "assign_half" in r["Name"] or "split_rows" in r["Name"]:
        print("ABL=$a", r["Name"][:40], "avg ms", round(float(r["AverageNs"])/1e6,4))
PY
done
