"""Exact-queue statistics per iteration (debug build: make -C hsg_amd/csrc EXTRA=-DHSGK_Q_STATS): entries by
candidate count (1 .. 5, 6-7, all-K)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
from hsg_amd.utils.segsort import common as sc
L = _lib.lib()
SHAPES = {'cfg4': (4, 256, 768, 768, [16, 16]), 'cfg2': (48, 256, 448, 448, [8, 8]), 'cfg5': (24, 384, 224, 224, [8, 16]),
          'cfg3': (16, 256, 224, 224, [8, 8])}
B, C, H, W, grid = SHAPES[sys.argv[1] if len(sys.argv) > 1 else 'cfg4']
flavour = sys.argv[2] if len(sys.argv) > 2 else 'iid'
from hsg_amd.utils import synth
x = synth.device_embeddings_nchw(synth.SEED_BASE + 4, (B, C, H, W), flavour, 'cuda:0')
print('qstats', sys.argv[1:], flush=True)
out = (ctypes.c_ulonglong * 8)()
prev = [0] * 8
for it in range(1, 11):
  sc.segment_by_kmeans(x, None, grid, iterations=it); torch.cuda.synchronize()
  L.hsgk_debug_qstats(out)
  cur = list(out)
  print('iterations 1..%d: entries with 1 / 2 / 3 / 4 / 5 / 6-7 candidates, all-K: %s' % (it, cur[1:8]), ' of', B * H * W, 'rows per iteration')
