#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   bash tools/collect_profiles.sh r02   -> gpurun_out/profiles/r02_*
# Counters are collected in their own passes (--kernel-trace + --pmc only).
set -u
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $out/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- \
  python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
cp /tmp/prof_k/k_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
{
  echo "# rocprofv3 --kernel-trace --pmc <counters>, one pass per line, python bench.py --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra (cfg2); means per dispatch"
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
             "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"; do
    rm -rf /tmp/prof_p
    rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_p -o p -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
    echo "## --pmc $pmc"
    python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/prof_p/p_counter_collection.csv hsgk init_meta_kernel,build_tables_kernel,count_valid_kernel,table_kernel,scan_chained_kernel,relabel_begin_kernel,relabel_ranked_kernel,sum_qcount_kernel
  done
} > $out/${tag}_pmc.txt 2>&1
# training-resolution workload (the fused per-image Lloyd kernel) and the other per-GPU configs
rm -rf /tmp/prof_t
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- \
  python $GRAFT_REPO_ROOT/bench.py --workload train28 --steps 20 --warmup 3 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
cp /tmp/prof_t/t_kernel_stats.csv $out/${tag}_train28_kernel_stats.csv
rm -rf /tmp/prof_r
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o r -- \
  python $GRAFT_REPO_ROOT/bench.py --workload reftrain --steps 20 --warmup 3 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
cp /tmp/prof_r/r_kernel_stats.csv $out/${tag}_reftrain_kernel_stats.csv
for wl in train28 train14 reftrain cfg3 cfg4 cfg5; do
  python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 10 --warmup 3 --cpu-images 0 --no-extra 2>/dev/null | tail -1 > $out/${tag}_bench_${wl}.json
done
python $GRAFT_REPO_ROOT/tools/bench_ops.py 2>/dev/null | tail -1 > $out/${tag}_ops.json
ls -la $out
