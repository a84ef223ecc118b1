"""Two streams issuing small-map calls at the same time (the co-residency cap of lloyd_small_groups keeps
two multi-workgroup grids within the CUs): every result equal to the single-stream one, no timeouts.
  python tools/probes/two_stream_small.py [reps]      (HSGK_SMALL / HSGK_SMALL_GROUPS select the route)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
total_bad = 0
for shape, grid, iters in (((48, 256, 28, 28), [8, 8], 10), ((4, 128, 56, 56), [4, 4], 15), ((20, 128, 40, 40), [4, 4], 8)):
  xs = [torch.randn(shape, device=dev) for _ in range(2)]
  ref = [[t.clone() for t in sc.segment_by_kmeans(x, None, grid, iterations=iters)] for x in xs]
  torch.cuda.synchronize()
  streams = [torch.cuda.Stream() for _ in range(2)]
  bad = 0
  t0 = time.time()
  for rep in range(reps):
    outs = []
    for i, st in enumerate(streams):
      with torch.cuda.stream(st):
        # no label map: no host sync inside the call, so the two streams really overlap
        outs.append(sc.segment_by_kmeans(xs[i], None, grid, iterations=iters))
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, ref)):
      if not all(torch.equal(a, b) for a, b in zip(o, r)):
        bad += 1
        print('  rep %d stream %d differs:' % (rep, i), [int((a != b).sum()) for a, b in zip(o, r)])
  total_bad += bad
  print(shape, grid, '%d x 2 concurrent calls: %d differing results, %.3f ms per pair' % (reps, bad, (time.time() - t0) / reps * 1e3))
print('ok' if total_bad == 0 else 'MISMATCHES: %d' % total_bad)
