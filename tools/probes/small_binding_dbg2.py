"""Where a small-map parity test spends its wall time under each host binding: H2D, the operator, D2H, the oracle."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
from oracle import oracle
os.environ['HSGK_SMALL'] = '1'
os.environ['HSGK_BINDING'] = sys.argv[1]
dev = torch.device('cuda:0')
shape, grid, iters = (5, 256, 28, 28), (8, 8), 10
B, C, H, W = shape
x = synth.embeddings_nchw(synth.SEED_BASE + 3 * C + H, shape, 'iid')
lab = synth.overseg_labels(synth.SEED_BASE + 8, B, H, W, regions=4, ignore_rows=2)
loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
def T():
  torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
  for l, ign in ((lab, 255), (None, None)):
    t0 = T()
    xd = torch.from_numpy(x).to(dev); ld = None if l is None else torch.from_numpy(l).to(dev)
    t1 = T()
    out = sc.segment_by_kmeans(xd, ld, list(grid), ignore_index=ign, iterations=iters)
    t2 = T()
    got = [t.cpu().numpy() for t in out]
    t3 = T()
    ref = oracle.segment_by_kmeans(x, l, grid, loc, ign, iters)
    t4 = T()
    ok = all(np.array_equal(a, b) for a, b in zip(got, ref))
    print(sys.argv[1], rep, 'labels' if l is not None else 'nolabels', 'h2d %.3f op %.3f d2h %.3f oracle %.3f s' % (t1 - t0, t2 - t1, t3 - t2, t4 - t3), ok, flush=True)
