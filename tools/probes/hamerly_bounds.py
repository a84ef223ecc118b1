"""Round 6, review item 7: would Hamerly-style bounds let the E-step skip rows, on the inputs SURVEY 8(d) names --
i.i.d. AND the mixture flavour (where centroids stop moving)?  One image of cfg2 (448 x 448, K = 64) and of cfg4
(768 x 768, K = 256), ten Lloyd iterations of the library's own operators.

Per row a lower bound m on (best score - second best score) is kept the way a skipping kernel would keep it (one
float per row): after an E-step that scored the row, m = its exact margin; every later iteration lowers it by
|c_best' - c_best| + max_{k != best} |c_k' - c_k| (unit rows: a score moves by at most the centroid's movement).  A
row with m > 0 provably keeps its label and need not be read.  Printed per iteration: the fraction of rows with m > 0
(skippable), and the fraction whose label really changed.

    python tools/probes/hamerly_bounds.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hsg_amd.utils.segsort.common as sc          # noqa: E402
from hsg_amd.utils import synth                      # noqa: E402

dev = torch.device('cuda:0')
for tag, cid, (C, H, W), grid in (('cfg2', 2, (256, 448, 448), (8, 8)), ('cfg4', 4, (256, 768, 768), (16, 16))):
  K = grid[0] * grid[1]
  for flav in ('iid', 'mixture'):
    x = synth.device_embeddings_nchw(synth.SEED_BASE + cid, (1, C, H, W), flav, dev)
    rows = sc.segment_by_kmeans(x, None, list(grid), iterations=1)[1]
    init = sc.initialize_cluster_labels(list(grid), (H, W), dev).view(-1)
    init = torch.unique(init, return_inverse=True)[1]
    labels = [init]
    for t in range(1, 11):
      labels.append(sc.kmeans_with_initial_labels(rows, labels[-1], K, 1))
    cents = [sc.calculate_prototypes_from_labels(rows, labels[t], K) for t in range(10)]   # E-step t + 1 scores these
    m = torch.full((rows.shape[0],), -1.0, device=dev)       # no bound before the first E-step
    out = []
    ar = torch.arange(K, device=dev)[None, :]
    for t in range(10):                                       # E-step t + 1: labels[t] -> labels[t + 1]
      if t > 0:
        delta = (cents[t] - cents[t - 1]).norm(dim=1)
        best = labels[t]
        dother = torch.empty_like(m)
        for s in range(0, rows.shape[0], 1 << 17):           # (chunks: the [N, K] mask of a 768 x 768 image is large)
          b = best[s:s + (1 << 17)]
          dother[s:s + (1 << 17)] = torch.where(ar == b[:, None], torch.zeros((), device=dev),
                                                delta[None, :].expand(b.numel(), K)).max(dim=1).values
        m = m - (delta[best] + dother)
      skip = m > 1e-5
      changed = labels[t + 1] != labels[t]
      assert not bool((skip & changed).any()), 'a bound was wrong'
      # rows that are scored get a fresh exact margin
      sc_ = torch.empty_like(m)
      for s in range(0, rows.shape[0], 1 << 17):
        top2 = (rows[s:s + (1 << 17)] @ cents[t].t()).topk(2, dim=1).values
        sc_[s:s + (1 << 17)] = top2[:, 0] - top2[:, 1]
      m = torch.where(skip, m, sc_)
      out.append((float(skip.float().mean()), float(changed.float().mean())))
    print('%s %-7s skippable rows per E-step: %s' % (tag, flav, ' '.join('%.3f' % a for a, _ in out)))
    print('%s %-7s labels changed          : %s' % (tag, flav, ' '.join('%.3f' % b for _, b in out)))
