#!/usr/bin/env python3
"""Pins further oracle functions against the REAL reference on random inputs (build container only,
CPU): calculate_prototypes_from_labels, find_nearest_prototypes, prepare_prototype_labels,
find_majority_label_index, the SegSortLoss log-likelihood (both modes) and SetSegSortLoss.
Floats within 2e-6 (prototypes; 2.5e-7 x the condition of the sum when that is larger) / 1e-4 (per-pixel nll, pixels whose fp32 '+' numerator cancels
in the reference left out), integers identical (nearest prototype: unless the fp64 top-2 margin
is below 1e-6).  Summary appended to profiles/r01_oracle_vs_reference.txt.

  python tests/checkers/fuzz_oracle_ops_vs_reference.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import hsg.utils.segsort.common as ref_common         # noqa: E402
import hsg.utils.segsort.loss as ref_loss             # noqa: E402
from hsg_amd.utils import synth                       # noqa: E402
from oracle import oracle                              # noqa: E402

torch.set_num_threads(8)
T = torch.from_numpy


def protos(rng):
  n, d, P = int(rng.integers(1, 9000)), int(rng.integers(1, 300)), int(rng.integers(1, 400))
  seed = int(rng.integers(1, 1 << 30))
  x = synth.gaussish(seed, n * d).reshape(n, d)
  lab = (synth.hash_u64(seed + 1, n) % np.uint64(P)).astype(np.int64)
  ref = ref_common.calculate_prototypes_from_labels(T(x), T(lab), P).numpy()
  got = oracle.calculate_prototypes_from_labels(x, lab, P)
  err = float(np.abs(got - ref).max())
  # conditioning of a normalised fp32 sum: sum of the row norms over the norm of the sum
  nrm = np.linalg.norm(x.astype(np.float64), axis=1)
  acc = np.zeros((P, d)); np.add.at(acc, lab, x.astype(np.float64))
  tot = np.zeros(P); np.add.at(tot, lab, nrm)
  an = np.linalg.norm(acc, axis=1)
  cond = float((tot[an > 0] / an[an > 0]).max()) if (an > 0).any() else 1.0
  tol = max(2e-6, 2.5e-7 * cond)
  return 'prototypes n=%d d=%d P=%d max|d|=%.1e (tolerance %.1e: fp32 sums in two orders, condition %.0f)' % (
      n, d, P, err, tol, cond), err <= tol


def nearest(rng):
  n, d, K = int(rng.integers(1, 6000)), int(rng.integers(2, 300)), int(rng.integers(1, 300))
  seed = int(rng.integers(1, 1 << 30))
  x = oracle.normalize_embedding(synth.gaussish(seed, n * d).reshape(n, d))
  p = oracle.normalize_embedding(synth.gaussish(seed + 1, K * d).reshape(K, d))
  ref = ref_common.find_nearest_prototypes(T(x), T(p)).numpy()
  got = oracle.find_nearest_prototypes(x, p)
  diff = np.flatnonzero(ref != got)
  worst = 0.0
  for i in diff:
    sc = np.sort(p.astype(np.float64) @ x[i].astype(np.float64))
    worst = max(worst, float(sc[-1] - sc[-2]))
  return 'find_nearest n=%d d=%d K=%d: %d differ, fp64 margin %.1e' % (n, d, K, len(diff), worst), worst < 1e-6


def labels(rng):
  n = int(rng.integers(1, 5000))
  seed = int(rng.integers(1, 1 << 30))
  sem = (synth.hash_u64(seed, n) % np.uint64(int(rng.integers(1, 30)))).astype(np.int64)
  inst = (synth.hash_u64(seed + 1, n) % np.uint64(int(rng.integers(1, 40)))).astype(np.int64)
  r_sem, r_idx = ref_common.prepare_prototype_labels(T(sem), T(inst), 256)
  o_sem, o_idx = oracle.prepare_prototype_labels(sem, inst, 256)
  ok = np.array_equal(r_sem.numpy(), o_sem) and np.array_equal(r_idx.numpy(), o_idx)
  clu = (synth.hash_u64(seed + 2, n) % np.uint64(int(rng.integers(1, 50)))).astype(np.int64)
  r = ref_common.find_majority_label_index(T(sem), T(clu))
  o = oracle.find_majority_label_index(sem, clu)
  ok = ok and all(np.array_equal(a.numpy(), b) for a, b in zip(r, o))
  return 'prepare_prototype_labels + find_majority_label_index n=%d' % n, ok


def loss(rng):
  n, c, P = int(rng.integers(1, 6000)), int(rng.choice([16, 32, 64, 128, 256, int(rng.integers(2, 200))])), int(rng.integers(1, 300))
  nsem, kappa = int(rng.integers(1, 22)), int(rng.integers(4, 21))
  seed = int(rng.integers(1, 1 << 30))
  e = oracle.normalize_embedding(synth.gaussish(seed, n * c).reshape(n, c))
  p = oracle.normalize_embedding(synth.gaussish(seed + 1, P * c).reshape(P, c))
  inst = (synth.hash_u64(seed + 2, n) % np.uint64(P)).astype(np.int64)
  psem = (synth.hash_u64(seed + 3, P) % np.uint64(nsem)).astype(np.int64)
  sem = psem[inst]
  sims = np.exp(float(kappa) * (e.astype(np.float64) @ p.astype(np.float64).T))
  own = sims[np.arange(n), inst]
  same = (sims * (sem[:, None] == psem[None, :])).sum(1) - own
  well = ~((same > 0) & (own > 100.0 * same))
  ok, info = True, []
  for mode in ('segsort+', 'segsort'):
    ref = ref_loss._calculate_log_likelihood(T(e), T(sem), T(inst), T(p), T(psem), kappa, mode).view(-1).numpy()
    got = oracle.segsort_nll(e, sem, inst, p, psem, float(kappa), mode)
    sel = well if mode == 'segsort+' else np.ones(n, bool)
    err = float(np.abs(got - ref)[sel].max()) if sel.any() else 0.0
    dm = abs(float(got[sel].astype(np.float64).sum()) - float(ref[sel].astype(np.float64).sum())) / n
    info.append('%s |d loss| %.1e max|d nll| %.1e' % (mode, dm, err))
    ok = ok and err <= 1e-4 and dm <= 1e-4
  return 'segsort_nll n=%d c=%d P=%d kappa=%d (%d ill-conditioned px) %s' % (n, c, P, kappa, int((~well).sum()), '; '.join(info)), ok


def main():
  n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
  bad, t0 = 0, time.time()
  fns = (protos, nearest, labels, loss)
  for case in range(n_cases):
    name, ok = fns[case % len(fns)](rng)
    print('case %3d: %s  %s' % (case, name, 'agrees' if ok else 'DIFFERENT'), flush=True)
    bad += 0 if ok else 1
  print('%d of %d oracle-function cases agree with the reference (%.0f s)' % (n_cases - bad, n_cases, time.time() - t0))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
