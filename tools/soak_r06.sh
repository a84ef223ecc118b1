#!/bin/bash
# Round-6 soak at the final kernels (one gpurun call): full-size parity, randomised parity of k-means / operators / exchange / loss
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $out
cd $GRAFT_REPO_ROOT
{
echo "# round 6 soak at $(cat .soak_commit 2>/dev/null): tails of"
echo "# full_parity_cfg2.py 48 | fuzz_parity.py 1500 401 | HSGK_FUZZ_LARGE=1 120 402 | HSGK_FUZZ_EXTREME=1 200 403 | fuzz_ops.py 700 404 | fuzz_exchange.py 400 405 | fuzz_loss_bwd.py 500 406 | fuzz_small_groups.py"
timeout 900 python tests/checkers/full_parity_cfg2.py 48 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tests/checkers/fuzz_parity.py 1500 401 2>&1 | grep -v amdgpu | tail -2
HSGK_FUZZ_LARGE=1 timeout 600 python tests/checkers/fuzz_parity.py 120 402 2>&1 | grep -v amdgpu | tail -2
HSGK_FUZZ_EXTREME=1 timeout 600 python tests/checkers/fuzz_parity.py 200 403 2>&1 | grep -v amdgpu | tail -2
timeout 600 python tests/checkers/fuzz_ops.py 700 404 2>&1 | grep -v amdgpu | tail -2
timeout 600 python tests/checkers/fuzz_exchange.py 400 405 2>&1 | grep -v amdgpu | tail -1
timeout 600 python tests/checkers/fuzz_loss_bwd.py 500 406 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tests/checkers/fuzz_small_groups.py 2>&1 | grep -v amdgpu | tail -2
} > $out/r06_soak.txt 2>&1
cat $out/r06_soak.txt
