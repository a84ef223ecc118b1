#!/bin/bash
# Round 6, review item 1, step 0 (one gpurun call): what owns the un-overlapped time of prep_fast32_kernel?
#   (1) the kernel's memory side alone (tools/probes/prep_mem.hip) next to the plain stream mix (rw_mix.hip)
#   (2) counter passes for prep_fast32_kernel alone: VALU / VMEM / LDS activity, waits, wave levels, TA / TCC stalls
#   (3) its time in this box's bench (HIP events of the library's profiler)
# -> gpurun_out/profiles/r06_prep_step0.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
f=$out/r06_prep_step0.txt
cd /tmp && export TMPDIR=/tmp
{
  echo "# commit $(cat $GRAFT_REPO_ROOT/.profile_commit 2>/dev/null || echo unknown)"
  echo "## (1a) plain float4 streams (tools/probes/rw_mix.hip)"
  hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/probes/rw_mix.hip -o /tmp/rw 2>/dev/null && timeout 120 /tmp/rw
  echo "## (1b) the prep kernel's access pattern without arithmetic (tools/probes/prep_mem.hip)"
  hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/probes/prep_mem.hip -o /tmp/prep_mem 2>/dev/null && timeout 300 /tmp/prep_mem
  echo "## (3) prep / E / M / step of the cfg2 call (tools/probes/prep_time.py)"
  timeout 300 python $GRAFT_REPO_ROOT/tools/probes/prep_time.py 2>&1 | grep -v -i "amdgpu.ids\|warn"
  B="python $GRAFT_REPO_ROOT/bench.py"
  echo "## (2) rocprofv3 --kernel-trace --pmc <one group per pass>, bench.py --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra; means per dispatch"
  for pmc in \
    "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
    "SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
    "SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
    "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
    "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" \
    "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_WRITE_sum" \
    "FETCH_SIZE" "WRITE_SIZE"; do
    rm -rf /tmp/prof_p
    timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_p -o p -- $B --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
    echo "### --pmc $pmc"
    python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/prof_p/p_counter_collection.csv prep_fast32
  done
} > $f 2>&1
tail -5 $f
