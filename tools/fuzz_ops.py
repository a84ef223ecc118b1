"""Randomised parity of two building blocks against the CPU oracle, bit for bit:
  * segment_reduce (prototypes / means / raw sums, canonical order C2) on random (n, d, P) and id patterns;
  * the E-step C entry point (hsgk_lloyd_estep, all three filter settings) on random (B, HW, C, K) with
    exact ties, near ties at the scale of each filter's gap and zero centroids.
Not part of the test suite; output appended to profiles/r01_fuzz_parity.txt.

  python tools/fuzz_ops.py [n_cases] [seed]
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hsg_amd import _lib, ops                      # noqa: E402
from hsg_amd.utils import synth                   # noqa: E402
from oracle import oracle                          # noqa: E402  (checker only)


def seg_case(rng, dev):
  n = int(rng.integers(1, 40000))
  d = int(rng.choice([1, 2, 3, 7, 16, 34, 64, 66, 130, 256, 258, 386, 514, int(rng.integers(1, 520))]))
  P = int(rng.integers(1, min(3000, 4 * n + 1) + 1))
  seed = int(rng.integers(1, 1 << 30))
  x = oracle.normalize_embedding(synth.gaussish(seed, n * d).reshape(n, d))
  pat = int(rng.integers(0, 3))
  # (documented contract: the ids of one 2048-row chunk span <= 512 consecutive segments)
  P = min(P, 512) if pat != 0 else min(P, max(1, n // 5))
  if pat == 0:      # image-major style: monotone with jitter
    lab = np.clip((np.arange(n, dtype=np.int64) * P) // n + (synth.hash_u64(seed + 1, n) % np.uint64(5)).astype(np.int64) - 2, 0, P - 1)
  elif pat == 1:    # random over all segments
    lab = (synth.hash_u64(seed + 1, n) % np.uint64(P)).astype(np.int64)
  else:             # random over a few segments, most segments empty
    lab = (synth.hash_u64(seed + 1, n) % np.uint64(min(P, 5))).astype(np.int64) * max(1, P // 5)
    lab = np.minimum(lab, P - 1)
  if rng.integers(0, 2):
    lab[::int(rng.integers(2, 200))] = -1
  xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
  ok = True
  for mode in (0, 1, 2):
    got = ops.segment_reduce(xt, lt, P, mode).cpu().numpy()
    if mode == 0:
      ref = oracle.calculate_prototypes_from_labels(x, lab, P)
    else:
      ref = np.empty((P, d), np.float32)
      oracle.lib().orc_segment_sums(x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_int64(n), d,
                                    lab.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int64(P),
                                    oracle.CHUNK, ref.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
      if mode == 1:
        cnt = np.bincount(lab[lab >= 0], minlength=P).astype(np.float32)
        cnt[cnt == 0] = 1
        ref = ref / cnt[:, None]
    ok = ok and np.array_equal(got, ref)
  return 'segment_reduce n=%d d=%d P=%d pattern=%d' % (n, d, P, pat), ok


def estep_case(rng, dev):
  B = int(rng.integers(1, 4))
  HW = int(rng.integers(1, 6000))
  C = int(rng.choice([30, 32, 64, 126, 128, 192, 256, 320, 384, 448, int(rng.integers(2, 500))]))
  K = int(rng.choice([1, 2, 7, 37, 64, 65, 100, 128, 129, 200, 256, 257, 300, int(rng.integers(1, 320))]))
  D, n = C + 2, B * HW
  seed = int(rng.integers(1, 1 << 30))
  x = oracle.normalize_embedding(synth.gaussish(seed, n * D).reshape(n, D))
  cent = oracle.normalize_embedding(synth.gaussish(seed + 1, B * K * D).reshape(B * K, D)).reshape(B, K, D).copy()
  if K >= 8:
    cent[:, 3] = cent[:, 1]
    cent[:, 6] = oracle.normalize_embedding(cent[:, 2] + np.float32(3e-6) * cent[:, 5])
    cent[:, 0] = oracle.normalize_embedding(cent[:, 4] + np.float32(4e-4) * cent[:, 7])
    cent[:, K - 1] = 0.0
  L = _lib.lib()
  xt, ct = torch.from_numpy(x).to(dev), torch.from_numpy(cent).to(dev)
  wsb = L.hsgk_lloyd_workspace_bytes(B, HW, D, K)
  ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
  ref = np.concatenate([oracle.find_nearest_prototypes(x[b * HW:(b + 1) * HW], cent[b]) for b in range(B)])
  ok = True
  for unit in (2, 1, 0):
    out = torch.full((n,), -1, dtype=torch.int32, device=dev)
    _lib.check(L.hsgk_lloyd_estep(xt.data_ptr(), B, HW, D, K, ct.data_ptr(), out.data_ptr(), unit,
                                  ws.data_ptr(), wsb, _lib.stream_ptr()))
    ok = ok and np.array_equal(out.cpu().numpy().astype(np.int64), ref)
  return 'lloyd_estep B=%d HW=%d C=%d K=%d' % (B, HW, C, K), ok


def main():
  n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
  dev = torch.device('cuda:0')
  bad, t0 = 0, time.time()
  for case in range(n_cases):
    name, ok = (seg_case if case % 2 == 0 else estep_case)(rng, dev)
    print('case %3d: %-52s %s' % (case, name, 'identical' if ok else 'DIFFERENT'), flush=True)
    bad += 0 if ok else 1
  print('%d of %d operator cases bit-identical to the oracle (%.0f s)' % (n_cases - bad, n_cases, time.time() - t0))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
