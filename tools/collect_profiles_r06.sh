#!/bin/bash
# Round-6 rocprofv3 evidence (run on the GPU box through gpurun; every profiled command under `timeout`):
#   bash tools/collect_profiles_r06.sh  -> gpurun_out/profiles/r06_*
set -u
commit=$(cat $GRAFT_REPO_ROOT/.profile_commit 2>/dev/null || echo unknown)
tag=r06
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py"
timeout 600 $B --steps 10 --warmup 3 2>/dev/null | tail -1 > $out/${tag}_bench.json
rm -rf /tmp/prof_k
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- $B --steps 3 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
cp /tmp/prof_k/k_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
skip=init_meta_kernel,build_tables_kernel,count_valid_kernel,table_kernel,scan_chained_kernel,relabel_begin_kernel,relabel_ranked_kernel,sum_qcount_kernel
for wl in cfg2 cfg3 cfg4 cfg5; do
  {
    echo "# rocprofv3 --kernel-trace --pmc <counters>, one pass per line, python bench.py --workload $wl --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra; means per dispatch; commit $commit"
    for pmc in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"; do
      rm -rf /tmp/prof_p
      timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d /tmp/prof_p -o p -- $B --workload $wl --steps 2 --warmup 1 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
      echo "## --pmc $pmc"
      python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/prof_p/p_counter_collection.csv hsgk $skip
    done
  } > $out/${tag}_${wl}_pmc.txt 2>&1
done
for wl in cfg3 cfg4 cfg5; do
  rm -rf /tmp/prof_k; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- $B --workload $wl --steps 5 --warmup 2 --cpu-images 0 --no-exchange --no-extra > /dev/null 2>&1
  cp /tmp/prof_k/k_kernel_stats.csv $out/${tag}_${wl}_kernel_stats.csv
done
# the prototype exchange on the cfg2 output and the loss at N = 200704, C = 256, P = 3072 (both engines)
rm -rf /tmp/prof_k; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/tools/probes/exchange_run.py cfg2 > $out/${tag}_exchange_cfg2.txt 2>&1
cp /tmp/prof_k/k_kernel_stats.csv $out/${tag}_exchange_kernel_stats.csv
for eng in split fp32; do
  rm -rf /tmp/prof_k; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/tools/probes/loss_prof.py 200704 256 3072 $eng > /dev/null 2>&1
  cp /tmp/prof_k/k_kernel_stats.csv $out/${tag}_loss_${eng}_kernel_stats.csv
done
for wl in train28 train14 reftrain; do
  timeout 300 $B --workload $wl --steps 10 --warmup 3 --cpu-images 0 --no-extra 2>/dev/null | tail -1 > $out/${tag}_bench_${wl}.json
done
for wl in cfg3 cfg4 cfg5; do      # (with extra_runs: mixture, labelled + ignore band; cfg4: the hierarchy)
  timeout 400 $B --workload $wl --steps 10 --warmup 3 --cpu-images 0 2>/dev/null | tail -1 > $out/${tag}_bench_${wl}.json
done
for fl in iid mixture; do
  bash $GRAFT_REPO_ROOT/tools/probes/cfg_iter_trace.sh cfg4 $fl > $out/${tag}_cfg4_${fl}_iter_trace.txt 2>&1
done
bash $GRAFT_REPO_ROOT/tools/probes/cfg_iter_trace.sh cfg2 iid > $out/${tag}_cfg2_iter_trace.txt 2>&1
timeout 400 python $GRAFT_REPO_ROOT/tests/checkers/fuzz_exchange.py 300 2>&1 | grep -v amdgpu | tail -4 > $out/${tag}_fuzz_exchange.txt
timeout 600 python $GRAFT_REPO_ROOT/tools/bench_ops.py 2>/dev/null | tail -1 > $out/${tag}_ops.json
{ for p in train_step_wall train_step_gpu train_step_syncs train_step_gaps train_step_ctypes; do echo "== tools/probes/$p.py"; timeout 300 python -u $GRAFT_REPO_ROOT/tools/probes/$p.py 2>&1 | grep -v -i "amdgpu.ids\|warn"; done; } > $out/${tag}_train_step.txt
for shp in "200704 256 3072" "50176 256 1568" "9408 128 1536"; do timeout 200 python $GRAFT_REPO_ROOT/tools/probes/loss_time.py $shp 2>&1 | tail -1; done > $out/${tag}_loss_time.txt
bash $GRAFT_REPO_ROOT/tools/probes/loss_pmc.sh 200704 3072 > $out/${tag}_loss_pmc.txt 2>&1
ls -la $out
