#!/bin/bash
# Round 6, second GPU call: GPU suite at the advisor fixes, the extended memory-side probe of the prep kernel, the
# Hamerly-bound measurement (review item 7), same-box baselines for the small-batch and training-step items.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/profiles
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/r06_gputest_call2.txt
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/probes/prep_mem.hip -o /tmp/prep_mem 2>/dev/null && timeout 300 /tmp/prep_mem > $out/r06_prep_mem.txt 2>&1
timeout 600 python $GRAFT_REPO_ROOT/tools/probes/hamerly_bounds.py 2>&1 | grep -v -i "amdgpu.ids\|warn" > $out/r06_hamerly_bounds.txt
B="python $GRAFT_REPO_ROOT/bench.py"
for wl in cfg3 cfg5; do
  timeout 400 $B --workload $wl --steps 10 --warmup 3 --cpu-images 0 --no-extra 2>/dev/null | tail -1 > $out/r06_base_bench_${wl}.json
  bash $GRAFT_REPO_ROOT/tools/probes/cfg_iter_trace.sh $wl iid > $out/r06_base_${wl}_iter_trace.txt 2>&1
done
{ for p in train_step_wall train_step_gpu train_step_gaps; do echo "== tools/probes/$p.py"; timeout 300 python -u $GRAFT_REPO_ROOT/tools/probes/$p.py 2>&1 | grep -v -i "amdgpu.ids\|warn"; done; } > $out/r06_base_train_step.txt
cat $out/r06_gputest_call2.txt; cat $out/r06_prep_mem.txt | tail -20; cat $out/r06_hamerly_bounds.txt
