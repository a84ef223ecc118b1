cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python $GRAFT_REPO_ROOT/bench.py --workload ${1:-cfg4} --steps 3 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
head -8 /tmp/pk/k_kernel_stats.csv | cut -c1-150
rm -rf /tmp/pp; timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/pp -o p -- python $GRAFT_REPO_ROOT/bench.py --workload ${1:-cfg4} --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pp/p_counter_collection.csv assign_half
