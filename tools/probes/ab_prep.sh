# same-box A/B of the prep kernel: stream-mix reference, then rocprofv3 per-kernel time for each build flag
hipcc --offload-arch=gfx950 -O3 tools/probes/rw_mix.hip -o /tmp/rw 2>/dev/null; /tmp/rw | grep -A3 "grid 8192" | grep mix
for a in "$@"; do
  touch hsg_amd/csrc/prep.hip; make -C hsg_amd/csrc EXTRA="$a" -j8 > /dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ablp -o ab -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-exchange --cpu-images 0 > /dev/null 2>&1)
  python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/ablp/ab_kernel_stats.csv")):
    if "prep" in r["Name"]:
        print("[$a]", r["Name"][:30], "avg ms", round(float(r["AverageNs"])/1e6,4))
PY
done
