#!/bin/bash
# kernel stats + a PMC pass of the cfg4 workload (K = 256 E-step), summary to stdout
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/px /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/px -o x -- python $GRAFT_REPO_ROOT/bench.py --workload ${1:-cfg4} --steps 5 --warmup 2 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("/tmp/px/x_kernel_stats.csv")))
for r in rows[:6]: print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU --output-format csv -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/bench.py --workload ${1:-cfg4} --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pp/p_counter_collection.csv ${2:-wide1}
