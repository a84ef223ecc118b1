#!/bin/bash
# same-box A/B of the prep kernel's two scheduling experiments (HSGK_PREP_X bits: 2 = chain-wave issue priority,
# 4 = `embeddings` rows stored before the second chain), three interleaved rounds
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for o in 0 2 4 6; do
    echo "HSGK_PREP_X=$o  $(HSGK_PREP_X=$o timeout 300 python tools/probes/prep_time.py 2>&1 | grep prep | tail -1)"
  done
done
