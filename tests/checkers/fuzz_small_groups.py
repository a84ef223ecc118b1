"""Stress of the fused Lloyd kernel with several co-operating workgroups per image: random shapes with
513 .. 6000 rows per image, C in {128, 256}, K <= 64, with and without labels -- every output of the fused
route (HSGK_SMALL=1, the library's choice of workgroups per image or a forced one) against the per-kernel
route (HSGK_SMALL=0) of the same library, bit for bit.  Not part of the suite.

  python tests/checkers/fuzz_small_groups.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.utils.segsort import common as sc   # noqa: E402


def main():
  n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
  dev = torch.device('cuda:0')
  bad = 0
  t0 = time.time()
  for case in range(n_cases):
    B = int(rng.integers(1, 13))
    C = int(rng.choice([128, 256]))
    H = int(rng.integers(16, 80))
    W = int(rng.integers(max(8, 513 // H + 1), max(10, min(6000 // H, 100)) + 1))
    gy, gx = int(rng.integers(1, 9)), int(rng.integers(1, 9))
    iters = int(rng.integers(1, 16))
    groups = int(rng.choice([0, 0, 2, 3, 5, 7, 16]))
    x = torch.randn((B, C, H, W), device=dev, generator=torch.Generator(device=dev).manual_seed(int(rng.integers(1, 1 << 30))))
    lab = None
    ign = None
    if rng.integers(0, 2):
      lab = torch.randint(0, 7, (B, H, W), device=dev)
      lab[:, : int(rng.integers(0, 4))] = 255
      ign = 255
    outs = []
    for route in ('0', '1'):
      os.environ['HSGK_SMALL'] = route
      if route == '1' and groups:
        os.environ['HSGK_SMALL_GROUPS'] = str(groups)
      else:
        os.environ.pop('HSGK_SMALL_GROUPS', None)
      outs.append([t.clone() for t in sc.segment_by_kmeans(x, lab, [gy, gx], ignore_index=ign, iterations=iters)])
    same = all(torch.equal(a, b) for a, b in zip(*outs))
    bad += 0 if same else 1
    if not same or case % 25 == 24 or case == n_cases - 1:
      print('case %d: B=%d C=%d %dx%d grid %dx%d it=%d groups=%d labels=%d  %s'
            % (case, B, C, H, W, gy, gx, iters, groups, lab is not None, 'identical' if same else 'DIFFERENT'))
  print('%d of %d cases identical between the fused (multi-workgroup) and the per-kernel route (%.0f s)'
        % (n_cases - bad, n_cases, time.time() - t0))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
