#!/bin/bash
# rocprofv3 evidence for the other per-GPU configs (run on the GPU box through gpurun):
#   bash tools/collect_profiles_cfg.sh r03 cfg3 cfg4 cfg5  -> gpurun_out/profiles/r03_<cfg>_{kernel_stats.csv,pmc.txt}
# Counters are collected in their own passes (--kernel-trace + --pmc only).
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  rm -rf /tmp/prof_k
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- \
    python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 2 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
  cp /tmp/prof_k/k_kernel_stats.csv $out/${tag}_${wl}_kernel_stats.csv
  {
    echo "# rocprofv3 --kernel-trace --pmc <counters>, one pass per line, python bench.py --workload $wl --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra; means per dispatch"
    for pmc in "FETCH_SIZE" "WRITE_SIZE" \
               "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"; do
      rm -rf /tmp/prof_p
      rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_p -o p -- \
        python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
      echo "## --pmc $pmc"
      python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/prof_p/p_counter_collection.csv hsgk init_meta_kernel,build_tables_kernel,count_valid_kernel,table_kernel,scan_chained_kernel,relabel_begin_kernel,relabel_ranked_kernel,sum_qcount_kernel
    done
  } > $out/${tag}_${wl}_pmc.txt 2>&1
done
ls -la $out
