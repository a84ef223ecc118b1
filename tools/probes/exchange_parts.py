"""Where the local part of the prototype exchange (hsg_amd/models/utils.exchange_prototypes, one rank) spends
its time, at the benchmark shape and at the training resolution."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import ops
from hsg_amd.models import utils as mu
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')


def timed(fn, reps=5):
  fn(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    out = fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e3, out


for shape, grid in (((48, 256, 448, 448), [8, 8]), ((48, 256, 28, 28), [8, 8])):
  x = synth.device_embeddings_nchw(synth.SEED_BASE + 2, shape, 'iid', dev)
  emb, eloc, lab, cidx, bidx = sc.segment_by_kmeans(x, None, grid, iterations=10)
  zeros = torch.zeros_like(lab)
  del x
  t_all, res = timed(lambda: mu.gather_clustering_and_update_prototypes(emb, eloc, cidx, bidx, lab, zeros))
  c, b, sem, inst = cidx, bidx, lab, zeros

  def keys():
    rc = c.max() + 1
    rl = torch.maximum(inst.max(), sem.max()) + 1
    return ((b * rc + c) * rl + sem) * rl + inst
  t_keys, k = timed(keys)
  t_uni, (lk, ids) = timed(lambda: torch.unique(k, return_inverse=True))
  P = lk.shape[0]
  t_s1, _ = timed(lambda: ops.segment_reduce(emb, ids, P, 2))
  t_s2, _ = timed(lambda: ops.segment_reduce(eloc, ids, P, 2))
  print('%s: whole %.3f ms | keys %.3f  unique %.3f  sums(emb) %.3f  sums(emb_loc) %.3f  (P = %d, N = %d)'
        % (shape, t_all, t_keys, t_uni, t_s1, t_s2, P, emb.shape[0]))
