"""Phase breakdown of the prep kernel (debug build: make -C hsg_amd/csrc EXTRA=-DHSGK_PREP_TIMING):
cycle counters of thread 0 of every workgroup, summed per phase."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib                          # noqa: E402
from hsg_amd.utils.segsort import common as sc   # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
x = torch.randn((48, 256, 448, 448), device=dev)
out = (ctypes.c_ulonglong * 8)()
sc.segment_by_kmeans(x, None, [8, 8], iterations=1)
torch.cuda.synchronize()
L.hsgk_debug_prep_timing(out)
sc.segment_by_kmeans(x, None, [8, 8], iterations=1)
torch.cuda.synchronize()
L.hsgk_debug_prep_timing(out)
names = ['0 bookkeeping (wave 0)', '1 (empty-tile check)', '2 loads -> LDS', '3 chain 1', '4 divide', '5 chain 2',
         '6 row stores + sums', '7 partial write-out']
tot = sum(out)
nwg = 48 * 2 * (448 * 448 // 64)
for n, v in zip(names, out):
  print('%-26s %6.1f %%   %8.0f ticks per workgroup' % (n, 100.0 * v / tot, v / nwg))
print('total %.0f ticks per workgroup' % (tot / nwg))
