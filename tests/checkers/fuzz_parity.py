"""Randomised parity run: segment_by_kmeans through the C ABI against the CPU oracle on random
shapes / grids / iteration counts / label maps / input distributions, everything bit for bit
(both float outputs, labels, cluster ids, batch ids).  Not part of the test suite (minutes);
the committed output is profiles/r01_fuzz_parity.txt.

  python tests/checkers/fuzz_parity.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.utils import synth                   # noqa: E402
from hsg_amd.utils.segsort import common as sc   # noqa: E402
from oracle import oracle                          # noqa: E402  (checker only)


LARGE = os.environ.get('HSGK_FUZZ_LARGE') == '1'       # image sides 90..329: many chunks / passes per image
EXTREME = os.environ.get('HSGK_FUZZ_EXTREME') == '1'   # degenerate inputs instead of the two usual distributions


def main():
  n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20260928)
  dev = torch.device('cuda:0')
  bad = 0
  t0 = time.time()
  for case in range(n_cases):
    B = int(rng.integers(1, 4))
    C = int(rng.choice([32, 64, 128, 192, 256, 256, 320, 384, 30, 100]))
    H, W = (int(rng.integers(12, 90)), int(rng.integers(12, 90))) if not LARGE else \
        (int(rng.integers(90, 330)), int(rng.integers(90, 330)))
    gy, gx = int(rng.integers(1, min(H, 17))), int(rng.integers(1, min(W, 17)))
    iters = int(rng.integers(0, 9))
    kind = str(rng.choice(['iid', 'mixture'] if not EXTREME else ['zeros', 'const', 'tiny', 'huge', 'dup']))
    seed = int(rng.integers(1, 1 << 30))
    x = synth.embeddings_nchw(seed, (B, C, H, W), 'mixture' if kind == 'mixture' else 'iid')
    if kind == 'zeros':       # a third of the pixels are all-zero vectors (eps path of the first normalisation)
      x = x * (synth.hash_u64(seed + 5, B * H * W) % np.uint64(3) != 0).astype(np.float32).reshape(B, 1, H, W)
    elif kind == 'const':     # every pixel identical: all scores tie, location decides
      x = np.broadcast_to(x[:, :, :1, :1], x.shape).copy()
    elif kind == 'tiny':      # norms below eps
      x = x * np.float32(1e-30)
    elif kind == 'huge':      # squares near the top of the float32 range
      x = x * np.float32(3e17)
    elif kind == 'dup':       # pixel pairs with identical embeddings (exact score ties across neighbours)
      x[:, :, :, 1::2] = x[:, :, :, 0:-1:2][:, :, :, :x[:, :, :, 1::2].shape[3]]
    mode = int(rng.integers(0, 3))          # 0: no labels, 1: oversegmentation + ignored rows, 2: labels without ignore
    lab, ign = None, 255
    if mode == 1:
      lab = synth.overseg_labels(seed + 1, B, H, W, regions=int(rng.integers(2, 9)),
                                 ignore_rows=int(rng.integers(0, 4)))
    elif mode == 2:
      lab = synth.overseg_labels(seed + 1, B, H, W, regions=int(rng.integers(2, 9)), ignore_rows=0)
    loc = oracle.generate_location_features((H, W)) - np.float32(0.5)     # the oracle's own restatement
    got = sc.segment_by_kmeans(torch.from_numpy(x).to(dev), None if lab is None else torch.from_numpy(lab).to(dev),
                               [gy, gx], ignore_index=ign, iterations=iters)
    got = [t.cpu().numpy() for t in got]
    ref = oracle.segment_by_kmeans(x, lab, (gy, gx), loc, ign, iters)
    ok = all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got, ref))
    print('case %3d: B=%d C=%3d %2dx%2d grid %2dx%2d it=%d %-7s labels=%d seed=%d  %s' %
          (case, B, C, H, W, gy, gx, iters, kind, mode, seed, 'identical' if ok else 'DIFFERENT'), flush=True)
    if not ok:
      for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
        if a.shape != b.shape:
          print('    %s: shape %s vs %s' % (name, a.shape, b.shape))
        elif not np.array_equal(a, b):
          d = np.argwhere(a != b)
          print('    %s: %d of %d elements differ, first at %s: %r vs %r' %
                (name, len(d), a.size, d[0].tolist(), a[tuple(d[0])], b[tuple(d[0])]))
    bad += 0 if ok else 1
  print('%d of %d cases bit-identical to the oracle (%.0f s)' % (n_cases - bad, n_cases, time.time() - t0))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
