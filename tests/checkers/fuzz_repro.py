"""Re-run given fuzz cases (B C H W gy gx iters kind mode seed) and report mismatching cluster ids."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.utils import synth                   # noqa: E402
from hsg_amd.utils.segsort import common as sc   # noqa: E402
from oracle import oracle                          # noqa: E402

CASES = [
    (1, 64, 79, 33, 12, 6, 2, 'iid', 0, 886861544),
    (1, 384, 45, 20, 14, 11, 6, 'mixture', 2, 708218441),
    (2, 64, 77, 72, 16, 13, 7, 'iid', 1, 776711247),
    (1, 256, 53, 14, 8, 2, 2, 'mixture', 0, 52142766),
    (1, 320, 85, 30, 4, 14, 1, 'iid', 1, 868003612),
]
dev = torch.device('cuda:0')
for (B, C, H, W, gy, gx, iters, kind, mode, seed) in CASES:
  x = synth.embeddings_nchw(seed, (B, C, H, W), kind)
  lab = None
  # (the fuzzer draws regions / ignore_rows from its rng; labels only select rows, try without)
  loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
  for it in range(0, iters + 1):
    got = sc.segment_by_kmeans(torch.from_numpy(x).to(dev), None, [gy, gx], ignore_index=255, iterations=it)
    got = [t.cpu().numpy() for t in got]
    ref = oracle.segment_by_kmeans(x, lab, (gy, gx), loc, 255, it)
    nd = int((got[3] != ref[3]).sum()) if got[3].shape == ref[3].shape else -1
    print('C=%d %dx%d grid %dx%d it=%d: %d cluster ids differ; distinct %d vs %d' %
          (C, H, W, gy, gx, it, nd, len(np.unique(got[3])), len(np.unique(ref[3]))), flush=True)
