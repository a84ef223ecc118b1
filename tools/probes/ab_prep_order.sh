#!/bin/bash
# same-box A/B of prep_fast32_kernel's workgroup order (HSGK_PREP_ORDER=0: ids as they come, default: one contiguous
# eighth of an image per XCD): prep / E / M / step by the library's own HIP events, three interleaved pairs
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for o in 0 1; do
    echo "HSGK_PREP_ORDER=$o  $(HSGK_PREP_ORDER=$o timeout 300 python tools/probes/prep_time.py 2>&1 | grep prep | tail -1)"
  done
done
