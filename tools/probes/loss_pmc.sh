# PMC counters of the loss kernels at (N, C, P) = ($1, 256, $2): means per dispatch
cd /tmp && export TMPDIR=/tmp
N=${1:-50176}; P=${2:-1568}
for pmc in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_WAVES"; do
  rm -rf /tmp/pp_l
  timeout 240 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/pp_l -o p -- python $GRAFT_REPO_ROOT/tools/probes/loss_prof.py $N 256 $P > /dev/null 2>&1
  echo "## --pmc $pmc"
  python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pp_l/p_counter_collection.csv loss_ 0
done
