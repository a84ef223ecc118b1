"""Times the prototype exchange on the output of a k-means call (run under rocprofv3 for the kernel split):
python tools/probes/exchange_run.py [workload] [labels]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
from hsg_amd.models import utils as mu

wl = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
cfg_id, B, C, H, W, grid, iters = bench.WORKLOADS[wl]
dev = torch.device('cuda:0')
x = synth.device_embeddings_nchw(synth.SEED_BASE + cfg_id, (B, C, H, W), 'iid', dev)
lab = None
if len(sys.argv) > 2:
  lab = torch.from_numpy(synth.overseg_labels(5, B, H, W, regions=48, ignore_rows=4, ignore_index=255)).to(dev)
out = sc.segment_by_kmeans(x, lab, list(grid), ignore_index=255 if lab is not None else None, iterations=iters)
del x
emb, eloc, labels, cidx, bidx = out
zeros = torch.zeros_like(labels)
for i in range(5):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  res = mu.gather_clustering_and_update_prototypes(emb, eloc, cidx, bidx, labels, zeros)
  torch.cuda.synchronize()
  print('%s exchange %.3f ms, %d segments' % (wl, (time.perf_counter() - t0) * 1e3, res[0].shape[0]))
  del res
