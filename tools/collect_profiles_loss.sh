#!/bin/bash
# Loss evidence of round 3 after the backward rewrite (gpurun; every profiled command under `timeout`):
#   bash tools/collect_profiles_loss.sh  -> gpurun_out/profiles/r03_loss_*, r03_ops.json, r03_train_step.txt
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for eng in split fp32; do
  rm -rf /tmp/prof_k; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/tools/probes/loss_prof.py 200704 256 3072 $eng > /dev/null 2>&1
  cp /tmp/prof_k/k_kernel_stats.csv $out/r03_loss_${eng}_kernel_stats.csv
done
rm -rf /tmp/prof_k; HSGK_LOSS_BWD=mixed timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/tools/probes/loss_prof.py 200704 256 3072 split > /dev/null 2>&1
cp /tmp/prof_k/k_kernel_stats.csv $out/r03_loss_mixed_kernel_stats.csv
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python tools/probes/loss_prof.py 200704 256 3072 (tools/probes/loss_pmc.sh): means per dispatch"
  bash $GRAFT_REPO_ROOT/tools/probes/loss_pmc.sh 200704 3072; } > $out/r03_loss_pmc.txt 2>&1
for shape in "200704 256 3072" "50176 256 1568" "37632 256 3072" "9408 128 1536"; do
  timeout 120 python $GRAFT_REPO_ROOT/tools/probes/loss_time.py $shape 2>&1 | tail -1
done > $out/r03_loss_times.txt
timeout 600 python $GRAFT_REPO_ROOT/tools/bench_ops.py 2>/dev/null | tail -1 > $out/r03_ops.json
timeout 300 python $GRAFT_REPO_ROOT/tools/probes/train_step_time.py 2>&1 | grep -v amdgpu.ids > $out/r03_train_step.txt
ls -la $out
