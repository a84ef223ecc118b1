# same-box comparison: reference streaming probe, then the fp16 filter kernel (rocprofv3 per-kernel time)
hipcc --offload-arch=gfx950 -O3 tools/probes/read_patterns2.hip -o /tmp/rp2 2>/dev/null; /tmp/rp2 | grep "RB  528" | grep "contiguous range per WG, 8 waves, 256"
for a in "$@"; do
  touch hsg_amd/csrc/kmeans.hip; make -C hsg_amd/csrc EXTRA="$a" -j8 > /dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ablh -o ab -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --reps 5 --only em --unit 2 > /dev/null 2>&1)
  python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ablh/ab_kernel_stats.csv")):
    if "assign_half" in r["Name"] or "split_rows" in r["Name"] or "accumulate" in r["Name"]:
        print("[$a]", r["Name"][:44], "avg ms", round(float(r["AverageNs"])/1e6,4))
PY
done
