"""Phase breakdown of the fused small-map Lloyd kernel (debug build: make -C hsg_amd/csrc
EXTRA=-DHSGK_SMALL_TIMING): cycle counter of thread 0 of workgroup 0, summed per phase over a call."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
from hsg_amd.utils.segsort import common as sc
L = _lib.lib()
dev = torch.device('cuda:0')
for shape, grid, iters in (((48, 256, 28, 28), [8, 8], 10), ((16, 256, 14, 14), [8, 8], 10), ((4, 128, 56, 56), [4, 4], 15)):
  x = torch.randn(shape, device=dev)
  out = (ctypes.c_ulonglong * 12)()
  for _ in range(2):
    sc.segment_by_kmeans(x, None, grid, iterations=iters)
  torch.cuda.synchronize()
  L.hsgk_debug_small_timing(out)
  sc.segment_by_kmeans(x, None, grid, iterations=iters)
  torch.cuda.synchronize()
  L.hsgk_debug_small_timing(out)
  names = ['M update', 'flush + fp32', 'F chain + divide', 'E filter', 'X exact chains', '  E: staging', '  E: tiles', '  E: drain', '  fold: wait readers', '  fold: atomics', '  fold: wait writers', '  fold: read sums']
  tot = sum(out[:5])
  print(shape, grid, 'total %.1f us (100 MHz counter) for %d iterations' % (tot / 100.0, iters))
  for nme, v in zip(names, out):
    print('   %-18s %7.1f us  %5.1f %%' % (nme, v / 100.0, 100.0 * v / max(tot, 1)))
