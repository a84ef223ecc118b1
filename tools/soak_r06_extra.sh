#!/bin/bash
# Round-6 extra soak of the rebuilt prep kernel (other seeds than tools/soak_r06.sh): randomised segment_by_kmeans
# parity incl. label maps / ignore bands / odd sizes, large shapes, extreme inputs; both prep switches on a subset
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/profiles; mkdir -p $out
cd $GRAFT_REPO_ROOT
{
echo "# round 6 extra soak at $(cat .soak_commit 2>/dev/null)"
timeout 900 python tests/checkers/fuzz_parity.py 3000 611 2>&1 | grep -v amdgpu | tail -1
HSGK_FUZZ_LARGE=1 timeout 900 python tests/checkers/fuzz_parity.py 200 612 2>&1 | grep -v amdgpu | tail -1
HSGK_FUZZ_EXTREME=1 timeout 600 python tests/checkers/fuzz_parity.py 400 613 2>&1 | grep -v amdgpu | tail -1
HSGK_PREP_FLAT=0 timeout 600 python tests/checkers/fuzz_parity.py 500 614 2>&1 | grep -v amdgpu | tail -1
HSGK_PREP_ORDER=0 HSGK_PREP_X=6 timeout 600 python tests/checkers/fuzz_parity.py 500 615 2>&1 | grep -v amdgpu | tail -1
} > $out/r06_soak_extra.txt 2>&1
cat $out/r06_soak_extra.txt
