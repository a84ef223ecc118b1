#!/bin/bash
# same-box A/B of prep_fast32_kernel's phase 3 (HSGK_PREP_FLAT=0: rounds 2-5, default: rows in registers, emb_loc
# through LDS as 16-byte pieces), both with the XCD-contiguous order; three interleaved pairs
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for o in 0 1; do
    echo "HSGK_PREP_FLAT=$o  $(HSGK_PREP_FLAT=$o timeout 300 python tools/probes/prep_time.py 2>&1 | grep prep | tail -1)"
  done
done
