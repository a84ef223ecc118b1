"""calculate_prototypes_from_labels on the cfg2 k-means output (9.63 M rows x 256 / 258, 3 072 segments): ms per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
cfg_id, B, C, H, W, grid, iters = bench.WORKLOADS['cfg2']
dev = torch.device('cuda:0')
x = synth.device_embeddings_nchw(synth.SEED_BASE + cfg_id, (B, C, H, W), 'iid', dev)
emb, eloc, labels, cidx, bidx = sc.segment_by_kmeans(x, None, list(grid), iterations=iters)
del x
P = int(cidx.max()) + 1
for name, rows in (('embeddings', emb), ('embeddings_with_loc', eloc)):
  for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    p = sc.calculate_prototypes_from_labels(rows, cidx, P)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
  print('%s: %.3f ms per call, %.2f TB/s' % (name, dt * 1e3, rows.numel() * 4 / dt / 1e12))
