"""Randomised parity run of the prototype exchange (hsg_amd/csrc/exchange.hip through the list API of
hsg_amd/models/utils.py) against oracle.exchange_prototypes: ids, labels exact, both float tables bit for
bit.  Sources sit on different devices when the box has more than one GPU (the list API's cross-device copies of
tuple blocks and sum tables).  Not part of the test suite; output committed as profiles/r04_fuzz_exchange.txt.

  python tests/checkers/fuzz_exchange.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.models import utils as mu          # noqa: E402
from hsg_amd.utils import synth                  # noqa: E402
from oracle import oracle                        # noqa: E402  (checker only)


def main():
  n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
  dev = torch.device('cuda:0')
  ndev = torch.cuda.device_count()
  bad = 0
  max_rows = 0                         # largest prototype table (segments) seen
  t0 = time.time()
  for case in range(n_cases):
    nsrc = int(rng.integers(1, 4))
    C = int(rng.choice([16, 32, 64, 128, 200, 256, 384, 20, 130]))
    nimg = int(rng.integers(1, 5))
    ncl = int(rng.choice([1, 4, 16, 64, 300]))
    nsem, ninst = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    sorted_rows = bool(rng.integers(0, 2))
    sparse = bool(rng.integers(0, 4) == 0)
    share_images = bool(rng.integers(0, 4) == 0)
    parts = []
    for g in range(nsrc):
      n = int(rng.choice([0, 1, 700, 2048, 2049, 5000, 12000, 30000])) if rng.integers(0, 3) else int(rng.integers(1, 9000))
      seed = int(rng.integers(1, 1 << 30))
      e = synth.gaussish(seed, n * C).reshape(n, C).astype(np.float64)
      e = (e / np.maximum(np.sqrt((e * e).sum(1, keepdims=True)), 1e-30)).astype(np.float32)
      l = synth.gaussish(seed + 1, n * 2).reshape(n, 2) * np.float32(0.3)
      el = np.concatenate([e, l], 1).astype(np.float64)
      el = (el / np.sqrt((el * el).sum(1, keepdims=True) + 1e-300)).astype(np.float32)
      img = (synth.hash_u64(seed + 2, n) % np.uint64(nimg)).astype(np.int64)
      if sorted_rows:
        img = np.sort(img)
      cl = (synth.hash_u64(seed + 3, n) % np.uint64(ncl)).astype(np.int64)
      if sparse:
        cl = cl * 977 + 13
      parts.append(dict(emb=e, emb_loc=el, cluster=cl, batch=img + (0 if share_images else nimg * g),
                        sem=(synth.hash_u64(seed + 4, n) % np.uint64(nsem)).astype(np.int64),
                        inst=(synth.hash_u64(seed + 5, n) % np.uint64(ninst)).astype(np.int64)))
    if sum(p['emb'].shape[0] for p in parts) == 0:
      continue
    want = oracle.exchange_prototypes(parts)
    spread = ndev > 1 and bool(rng.integers(0, 2))
    devs = [torch.device('cuda', g % ndev) if spread else dev for g in range(nsrc)]
    T = lambda k: [torch.from_numpy(p[k]).to(devs[g]) for g, p in enumerate(parts)]
    max_rows = max(max_rows, int(want[2].shape[0]))
    got = mu.gather_clustering_and_update_prototypes(T('emb'), T('emb_loc'), T('cluster'), T('batch'), T('sem'), T('inst'), dev)
    ok = all(np.array_equal(got[j][0].cpu().numpy(), want[j]) for j in (2, 3, 4))
    ok = ok and all(np.array_equal(got[5][g].cpu().numpy(), want[5][g]) for g in range(nsrc))
    ok = ok and np.array_equal(got[0][0].cpu().numpy().view(np.uint32), want[0].view(np.uint32))
    ok = ok and np.array_equal(got[1][0].cpu().numpy().view(np.uint32), want[1].view(np.uint32))
    if not ok:
      bad += 1
      print('MISMATCH case %d: nsrc=%d C=%d nimg=%d ncl=%d sizes=%s sorted=%s sparse=%s share=%s'
            % (case, nsrc, C, nimg, ncl, [p['emb'].shape[0] for p in parts], sorted_rows, sparse, share_images), flush=True)
    if (case + 1) % 20 == 0:
      print('%d cases, %d mismatching, %.0f s' % (case + 1, bad, time.time() - t0), flush=True)
  print('fuzz_exchange: %d cases, %d mismatching (prototype tables of up to %d segments, %d device(s))'
        % (n_cases, bad, max_rows, ndev))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
