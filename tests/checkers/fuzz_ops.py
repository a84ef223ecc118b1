"""Randomised parity of two building blocks against the CPU oracle, bit for bit:
  * segment_reduce (prototypes / means / raw sums, canonical order C2) on random (n, d, P) and id patterns;
  * the E-step C entry point (hsgk_lloyd_estep, all three filter settings) on random (B, HW, C, K) with
    exact ties, near ties at the scale of each filter's gap and zero centroids;
  * SegSortLoss forward (both modes, random n / c / P / concentration): per-pixel nll and mean within 1e-4;
  * the hierarchy operators (grouping from logits, masked group means, pixel label lookup): labels identical,
    probabilities within 1e-6, means within 1e-5.
Not part of the test suite; output appended to profiles/r01_fuzz_parity.txt.

  python tests/checkers/fuzz_ops.py [n_cases] [seed]
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
  if K >= 24 and rng.integers(0, 2):        # a cloud of near copies of one centroid: 4..7 and more candidates
    for i, kk in enumerate((9, 10, 11, 12, 13)):
      cent[:, kk] = oracle.normalize_embedding(cent[:, 8] + np.float32(1e-5 * (i + 1)) * cent[:, 14 + i])
    if K > 140:
      for i, kk in enumerate((130, 131, 133, 139)):
        cent[:, kk] = oracle.normalize_embedding(cent[:, 8] + np.float32(1.5e-5 * (i + 1)) * cent[:, 20 + i])
    x[::17] = oracle.normalize_embedding(cent[0, 8][None, :] + np.float32(0.02) * x[::17])
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


def loss_case(rng, dev):
  """SegSortLoss forward (both modes): the loss and every per-pixel nll within the north_star
  tolerance 1e-4.  In 'segsort+' mode the reference forms the numerator as (sum over same-semantic
  prototypes) - (own prototype) in fp32 (loss.py:63-66), which cancels when the own term dominates:
  those pixels are summation-order noise in the reference itself (oracle and kernel sum in different
  orders) and are left out of the '+' comparison (counted in the output)."""
  from hsg_amd.utils.segsort.loss import SegSortLoss
  n = int(rng.integers(1, 20000))
  c = int(rng.choice([16, 32, 48, 64, 128, 256, int(rng.integers(2, 300))]))
  P = int(rng.integers(1, 600))
  nsem = int(rng.integers(1, 22))
  kappa = int(rng.integers(4, 21))
  seed = int(rng.integers(1, 1 << 30))
  e = oracle.normalize_embedding(synth.gaussish(seed, n * c).reshape(n, c))
  p = oracle.normalize_embedding(synth.gaussish(seed + 1, P * c).reshape(P, c))
  inst = (synth.hash_u64(seed + 2, n) % np.uint64(P)).astype(np.int64)
  psem = (synth.hash_u64(seed + 3, P) % np.uint64(nsem)).astype(np.int64)
  sem = psem[inst]
  t = lambda a: torch.from_numpy(a).to(dev)
  # condition of the reference's own '+' numerator (same-semantic sum minus own term, fp32): pixels
  # where the own term exceeds 100 x the remainder are summation-order noise in the reference
  sims = np.exp(float(kappa) * (e.astype(np.float64) @ p.astype(np.float64).T))
  own = sims[np.arange(n), inst]
  same = (sims * (sem[:, None] == psem[None, :])).sum(1) - own
  well = ~((same > 0) & (own > 100.0 * same))
  ok, info = True, []
  for mode in ('segsort+', 'segsort'):
    nll = SegSortLoss(kappa, mode, reduction='none')(t(e), t(sem), t(inst), t(p), t(psem)).view(-1).cpu().numpy()
    ref = oracle.segsort_nll(e, sem, inst, p, psem, float(kappa), mode)
    sel = well if mode == 'segsort+' else np.ones(n, bool)
    err = float(np.abs(nll - ref)[sel].max()) if sel.any() else 0.0
    dm = abs(float(nll[sel].astype(np.float64).sum()) - float(ref[sel].astype(np.float64).sum())) / n
    info.append('%s: |d loss| %.1e, max |d nll| %.1e' % (mode, dm, err))
    ok = ok and dm <= 1e-4 and err <= 1e-4
  return 'segsort_loss n=%d c=%d P=%d kappa=%d (%d ill-conditioned px) %s' % (
      n, c, P, kappa, int((~well).sum()), '; '.join(info)), ok


def hier_case(rng, dev):
  """a12-a14: softmax / argmax / Bayes-chain grouping, masked group means, pixel label lookup."""
  from hsg_amd.models.embeddings import hierarchy as hz
  B, M = int(rng.integers(1, 7)), int(rng.choice([16, 64, 256, int(rng.integers(2, 300))]))
  KF, KC = int(rng.integers(2, 70)), int(rng.integers(1, 20))
  C = int(rng.choice([16, 64, 256, int(rng.integers(2, 300))]))
  seed = int(rng.integers(1, 1 << 30))
  fl = (synth.gaussish(seed, B * KF * M).reshape(B, KF, M) * np.float32(2)).astype(np.float32)
  cl = (synth.gaussish(seed + 1, B * KC * KF).reshape(B, KC, KF) * np.float32(2)).astype(np.float32)
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  f_lab, f_prob, c_lab, c_prob = hz.hierarchical_grouping_from_logits(t(fl), t(cl))
  r = oracle.hierarchical_grouping_from_logits(fl, cl)
  ok = np.array_equal(f_lab.cpu().numpy(), r[0]) and np.array_equal(c_lab.cpu().numpy(), r[2])
  ok = ok and float(np.abs(f_prob.cpu().numpy() - r[1]).max()) <= 1e-6 and float(np.abs(c_prob.cpu().numpy() - r[3]).max()) <= 1e-6
  protos = synth.gaussish(seed + 2, B * C * M).reshape(B, C, M)
  masks = (synth.hash_u64(seed + 3, B * M) % np.uint64(5) == 0).reshape(B, M)
  for normalized in (False, True):
    got = hz.collect_nd_coarser_prototype(t(protos), t(r[0]), t(masks), KF, normalized).cpu().numpy()
    ref = oracle.collect_nd_coarser_prototype(protos, r[0], masks, KF, normalized)
    ok = ok and float(np.abs(got - ref).max()) <= 1e-5
  n = int(rng.integers(1, 20000))
  bidx = np.sort((synth.hash_u64(seed + 4, n) % np.uint64(B)).astype(np.int64)) * 3 + 5      # sparse image ids
  if len(np.unique(bidx)) == B:
    cby = (synth.hash_u64(seed + 5, n) % np.uint64(M)).astype(np.int64)
    got = hz.collect_pixel_hierarchical_clustering_indices(t(cby), t(bidx), t(r[0])).cpu().numpy()
    ok = ok and np.array_equal(got, oracle.collect_pixel_hierarchical_clustering_indices(cby, bidx, r[0]))
  return 'hierarchy B=%d M=%d KF=%d KC=%d C=%d' % (B, M, KF, KC, C), ok


def topk_case(rng, dev):
  """n1: top-k retrieval (indices wherever the score gaps exceed fp32 noise, values to rounding) against
  torch mm + argsort on the GPU, and the majority labels of clusters against the oracle."""
  from hsg_amd.utils.segsort import eval as ev
  from hsg_amd.utils.segsort import common as sc
  n, c, P = int(rng.integers(1, 6000)), int(rng.choice([16, 48, 128, 256, int(rng.integers(2, 300))])), int(rng.integers(1, 1500))
  k = int(rng.integers(1, min(P, 24) + 1))
  seed = int(rng.integers(1, 1 << 30))
  e = torch.from_numpy(oracle.normalize_embedding(synth.gaussish(seed, n * c).reshape(n, c))).to(dev)
  p = torch.from_numpy(oracle.normalize_embedding(synth.gaussish(seed + 1, P * c).reshape(P, c))).to(dev)
  got_idx, got_val = ev.top_k_indices(e, p, k)
  aff = torch.mm(e, p.t())
  srt, idx = torch.sort(aff, 1, descending=True)
  ok = float((got_val - srt[:, :k]).abs().max()) <= 2e-6
  gap_ok = torch.ones((n, k), dtype=torch.bool, device=dev)
  if P > k:
    gap_ok &= (srt[:, :k] - srt[:, 1:k + 1]) > 1e-5
  if k > 1:
    gap_ok[:, 1:] &= (srt[:, :k - 1] - srt[:, 1:k]) > 1e-5
  ok = ok and bool(torch.equal(got_idx[gap_ok], idx[:, :k][gap_ok]))
  sem = (synth.hash_u64(seed + 2, n) % np.uint64(int(rng.integers(1, 22)))).astype(np.int64)
  clu = (synth.hash_u64(seed + 3, n) % np.uint64(int(rng.integers(1, 80)))).astype(np.int64)
  sel, maj = sc.find_majority_label_index(torch.from_numpy(sem).to(dev), torch.from_numpy(clu).to(dev))
  r_sel, r_maj = oracle.find_majority_label_index(sem, clu)
  ok = ok and np.array_equal(sel.cpu().numpy(), r_sel) and np.array_equal(maj.cpu().numpy(), r_maj)
  return 'top_k n=%d c=%d P=%d k=%d + majority labels' % (n, c, P, k), ok


def main():
  n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
  dev = torch.device('cuda:0')
  bad, t0 = 0, time.time()
  for case in range(n_cases):
    name, ok = (seg_case, estep_case, loss_case, hier_case, topk_case)[case % 5](rng, dev)
    print('case %3d: %-60s  %s' % (case, name, ('within 1e-4' if name.startswith('segsort_loss') else 'identical') if ok else 'DIFFERENT'), flush=True)
    bad += 0 if ok else 1
  print('%d of %d operator cases agree with the oracle (bit-identical; loss: within 1e-4) (%.0f s)' % (n_cases - bad, n_cases, time.time() - t0))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
