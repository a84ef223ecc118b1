#!/bin/bash
# Rebuilds prep.o with experiment switches on the GPU box and times the prep kernel of the cfg2 call
# (HIP events of the library's profiler):  bash tools/probes/prep_variants.sh "" -DHSGK_PREP_NT ...
cd $GRAFT_REPO_ROOT
for flags in "$@"; do
  touch hsg_amd/csrc/prep.hip
  make -s -C hsg_amd/csrc EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  python - "$flags" <<'PY'
import sys, torch
sys.path.insert(0, '.')
from hsg_amd import _lib
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
x = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (48, 256, 448, 448), 'iid', dev)
for _ in range(2):
  sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
torch.cuda.synchronize()
_lib.profile_enable(True); _lib.profile_collect()
import time
t0 = time.perf_counter()
for _ in range(5):
  sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
p = _lib.profile_collect()
print('%-40s prep %.3f ms  step %.2f ms' % (sys.argv[1] or '(baseline)', p['prep'][0] / p['prep'][1], dt * 1e3))
PY
done
touch hsg_amd/csrc/prep.hip; make -s -C hsg_amd/csrc > /dev/null 2>&1
