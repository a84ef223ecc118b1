"""One-off full-size parity run: BASELINE.json configs[1] (48x256x448x448, K=8x8, 10 Lloyd
iterations, seeded i.i.d. input) through the C ABI, EVERY image compared bit for bit with the
CPU oracle (labels / cluster ids and both float outputs).  Prints one line per image and a
summary; the committed output is profiles/r01_full_parity.txt.

  python tests/checkers/full_parity_cfg2.py [n_images]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.utils.segsort import common as sc   # noqa: E402
from oracle import oracle                          # noqa: E402  (checker only)


def main():
  B, C, H, W, grid, iters = 48, 256, 448, 448, (8, 8), 10
  nimg = int(sys.argv[1]) if len(sys.argv) > 1 else B
  dev = torch.device('cuda:0')
  gen = torch.Generator(device=dev)
  gen.manual_seed(20260928)
  x = torch.randn((B, C, H, W), device=dev, generator=gen)
  emb, eloc, labels, cluster, batch = sc.segment_by_kmeans(x, None, list(grid), iterations=iters)
  torch.cuda.synchronize()
  loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
  bad = 0
  t0 = time.time()
  for b in range(nimg):
    ref = oracle.segment_by_kmeans(x[b:b + 1].cpu().numpy(), None, grid, loc, None, iters)
    sl = slice(b * H * W, (b + 1) * H * W)
    ok = (np.array_equal(emb[sl].cpu().numpy(), ref[0]), np.array_equal(eloc[sl].cpu().numpy(), ref[1]),
          np.array_equal((cluster[sl] - b * 64).cpu().numpy(), ref[3]))
    print('image %2d: emb %s  emb_loc %s  cluster ids %s  (%d distinct clusters)' %
          (b, 'identical' if ok[0] else 'DIFFERENT', 'identical' if ok[1] else 'DIFFERENT',
           'identical' if ok[2] else 'DIFFERENT', len(np.unique(ref[3]))), flush=True)
    bad += 0 if all(ok) else 1
  print('%d of %d images bit-identical to the oracle (%d pixels, %.0f s of oracle time)' %
        (nimg - bad, nimg, nimg * H * W, time.time() - t0))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
