# same-box A/B of one kernel: ab_kernel.sh <source file under hsg_amd/csrc> <kernel name substring> <build flag>...
# rebuilds with each flag (make EXTRA=flag) and prints the kernel's per-launch times from rocprofv3
src=$1; name=$2; shift 2
for a in "$@"; do
  touch hsg_amd/csrc/$src; make -C hsg_amd/csrc EXTRA="$a" -j8 > /dev/null 2>&1
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abk -o ab -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-exchange --cpu-images 0 > /dev/null 2>&1)
  python - <<PY
import csv
v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open("gpurun_out/abk/ab_kernel_trace.csv")) if "$name" in r["Kernel_Name"]]
n = len(v) // 4
print("[$a] $name: avg %.1f us; last step:" % (sum(v) / max(len(v), 1)), [round(x) for x in v[-n:]])
PY
done
