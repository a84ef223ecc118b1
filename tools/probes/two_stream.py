"""Would running the two halves of a small per-GPU batch on two streams hide the latency-bound kernels of one half's
Lloyd iteration (exact pass, sparse sums update, finalisation) behind the other half's filter?  The whole operator
on the full batch against two concurrent calls on its halves (their results differ from the full call's only in
the batch-wide dense ids; timing experiment)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
for tag, cid, shape, grid in (('cfg3', 3, (16, 256, 224, 224), [8, 8]), ('cfg5', 5, (24, 384, 224, 224), [8, 16]),
                              ('cfg4', 4, (4, 256, 768, 768), [16, 16]), ('cfg2/4', 2, (12, 256, 448, 448), [8, 8])):
  x = synth.device_embeddings_nchw(synth.SEED_BASE + cid, shape, 'iid', dev)
  B = shape[0]
  halves = [x[:B // 2].contiguous(), x[B // 2:].contiguous()]
  streams = [torch.cuda.Stream(), torch.cuda.Stream()]
  def full():
    return sc.segment_by_kmeans(x, None, grid, iterations=10)
  def split():
    outs = []
    for h, s in zip(halves, streams):
      with torch.cuda.stream(s):
        outs.append(sc.segment_by_kmeans(h, None, grid, iterations=10))
    return outs
  res = {}
  for name, fn in (('full', full), ('two streams', split), ('full', full), ('two streams', split)):
    for _ in range(3):
      fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
      fn()
    torch.cuda.synchronize()
    res.setdefault(name, []).append((time.perf_counter() - t0) / 10 * 1e3)
  print('%-7s %s: full batch %s ms, halves on two streams %s ms' %
        (tag, 'x'.join(map(str, shape)), ' / '.join('%.3f' % v for v in res['full']),
         ' / '.join('%.3f' % v for v in res['two streams'])), flush=True)
  del x, halves
