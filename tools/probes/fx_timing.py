"""Where a wave of the exact-sum update kernel spends its time (probe build: make EXTRA=-DHSGK_FX_TIMING; the
waits are made explicit in that build, so the kernel itself runs slower than the product's)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
L = _lib.lib()
dev = torch.device('cuda:0')
x = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (48, 256, 448, 448), 'iid', dev)
out = (ctypes.c_ulonglong * 8)()
names = ['labels + list of changed rows', 'waiting for a batch of rows (vmcnt)', 'conversion + issuing LDS atomics',
         'waiting for the atomics (lgkmcnt)', 'issuing row loads (entries in registers)', 'reading the list entries (LDS)']
prev = [0] * 8
sc.segment_by_kmeans(x, None, [8, 8], iterations=2); torch.cuda.synchronize(); L.hsgk_debug_fx_timing(out)
for iters in (2, 3, 4, 10):
  sc.segment_by_kmeans(x, None, [8, 8], iterations=iters); torch.cuda.synchronize()
  L.hsgk_debug_fx_timing(out)
  cur = list(out)
  d = [c - p for c, p in zip(cur, prev)] if iters > 2 else cur
  tot = sum(d[:6]) or 1
  print('update launches %s:' % ('1' if iters == 2 else '%d .. %d' % (1 if iters == 2 else {3: 2, 4: 3, 10: 4}[iters], iters - 1)))
  for n, v in zip(names, d[:6]):
    print('   %-40s %5.1f %%' % (n, 100.0 * v / tot))
  prev = cur
