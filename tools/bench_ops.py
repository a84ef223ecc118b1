#!/usr/bin/env python3
"""Secondary measurements (NOT the contract bench -- that is bench.py): the operators around
the k-means hot path at the shapes SURVEY.md 8(d) names, one JSON line with ms per call.
Everything goes Python mirror -> C ABI; wall time with torch.cuda.synchronize, best of 5.

  python tools/bench_ops.py            # on the GPU box
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def timeit(fn, reps=5):
  fn()
  torch.cuda.synchronize()
  best = 1e9
  for _ in range(reps):
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
  return round(best * 1e3, 3)


def main():
  from hsg_amd.models import utils as model_utils
  from hsg_amd.models.embeddings import hierarchy as hz
  from hsg_amd.utils.graph import common as gc
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort import eval as ev
  from hsg_amd.utils.segsort import inference as inf
  from hsg_amd.utils.segsort.loss import SegSortLoss
  dev = torch.device('cuda', 0)
  gen = torch.Generator(device=dev)
  gen.manual_seed(7)
  res = {}
  # one cfg2 image through k-means -> the loss / retrieval inputs (N = 200 704, C = 256)
  x = torch.randn((4, 256, 448, 448), device=dev, generator=gen)
  emb, emb_loc, lab, cidx, bidx = sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
  n1 = 448 * 448
  P = int(cidx.max().item()) + 1
  protos = sc.calculate_prototypes_from_labels(emb, cidx)
  psem = torch.arange(P, device=dev) % 21
  sem = psem[cidx]
  res['shapes'] = {'N_loss': n1, 'C': 256, 'P': P}
  e1, s1, c1 = emb[:n1], sem[:n1], cidx[:n1]
  res['segsort_loss_fwd_ms'] = timeit(lambda: SegSortLoss(16, 'segsort+')(e1, s1, c1, protos, psem))

  def fwd_bwd():
    e = e1.detach().requires_grad_(True)
    p = protos.detach().requires_grad_(True)
    SegSortLoss(16, 'segsort+')(e, s1, c1, p, psem).backward()
  res['segsort_loss_fwd_bwd_ms'] = timeit(fwd_bwd)
  res['segsort_loss_fwd_tflops'] = round(2.0 * 256 * P * n1 / (res['segsort_loss_fwd_ms'] * 1e-3) / 1e12, 1)
  # the loss where the reference runs it (predictions/hsg.py:105,130,149) and at one benchmark image against a
  # whole batch's table: (N, C, P); forward on the bf16x3 engine (default) and on the fp32 engine
  rows = {}
  for n_, c_, P_ in ((9408, 128, 1536), (37632, 256, 3072), (200704, 256, 3072)):
    pr = torch.nn.functional.normalize(torch.randn((P_, c_), device=dev, generator=gen), dim=1)
    ins = torch.randint(0, P_, (n_,), device=dev, generator=gen)
    ee = torch.nn.functional.normalize(pr[ins] + 0.35 * torch.randn((n_, c_), device=dev, generator=gen), dim=1)
    ps = torch.arange(P_, device=dev) % 21
    se = ps[ins]
    row = {}
    for eng in ('split', 'fp32'):
      os.environ['HSGK_LOSS'] = eng
      ms = timeit(lambda: SegSortLoss(16, 'segsort+')(ee, se, ins, pr, ps))
      row['fwd_ms_' + eng] = ms
      row['fwd_tflops_equiv_' + eng] = round(2.0 * c_ * P_ * n_ / (ms * 1e-3) / 1e12, 1)
    os.environ.pop('HSGK_LOSS', None)

    def fb():
      a = ee.detach().requires_grad_(True)
      b = pr.detach().requires_grad_(True)
      SegSortLoss(16, 'segsort+')(a, se, ins, b, ps).backward()
    row['fwd_bwd_ms'] = timeit(fb)
    row['bwd_tflops'] = round(8.0 * c_ * P_ * n_ / ((row['fwd_bwd_ms'] - row['fwd_ms_split']) * 1e-3) / 1e12, 1)
    rows['N%d_C%d_P%d' % (n_, c_, P_)] = row
  res['segsort_loss_scale'] = rows
  res['top_k_ranking_k20_ms'] = timeit(lambda: ev.top_k_ranking(e1, s1, protos, psem, 20))
  res['prototype_table_4img_ms'] = timeit(lambda: sc.calculate_prototypes_from_labels(emb, cidx))
  zeros = torch.zeros_like(lab)
  res['exchange_local_4img_ms'] = timeit(
      lambda: model_utils.gather_clustering_and_update_prototypes(emb, emb_loc, cidx, bidx, lab, zeros))
  sem_all = (torch.arange(emb.shape[0], device=dev) * 7919) % 21
  res['find_majority_label_index_ms'] = timeit(lambda: sc.find_majority_label_index(sem_all, cidx))
  del x, emb, emb_loc
  # hierarchy (cfg4: 256 -> 64 -> 16, B' = 32 image pairs)
  Bp, C, M, KF, KC = 32, 256, 256, 64, 16
  fl = torch.randn((Bp, KF, M), device=dev, generator=gen)
  cl = torch.randn((Bp, KC, KF), device=dev, generator=gen)
  res['hier_assign_ms'] = timeit(lambda: hz.hierarchical_grouping_from_logits(fl, cl))
  cen = torch.randn((Bp, C, KF), device=dev, generator=gen)
  nod = torch.randn((Bp, C, M), device=dev, generator=gen)
  res['transformer_clustering_tail_ms'] = timeit(lambda: hz.transformer_clustering_tail(cen, cen, nod, KF))
  xg = torch.nn.functional.normalize(nod, dim=1)
  pad = torch.zeros((Bp, M), dtype=torch.bool, device=dev)
  seg = (torch.arange(M, device=dev) % 2).expand(Bp, M).contiguous()
  res['knn_affinity_256nodes_knn10_ms'] = timeit(lambda: gc.affinity_matrix_as_attention(xg, pad, seg, 10))
  # overlap-averaged inference crops: 1024 x 2048 canvas, 512 x 512 crops, stride 384
  Hc, Wc, crop = 1024, 2048, 512
  crops = torch.randn((1, 256, crop, crop), device=dev, generator=gen)
  ends_h, ends_w = inf.patch_end_indices(Hc, crop, 384), inf.patch_end_indices(Wc, crop, 384)

  def overlap():
    avg = inf.OverlapAverager(256, Hc, Wc, dev)
    for eh in ends_h:
      for ew in ends_w:
        avg.add(crops, int(eh) - crop, int(ew) - crop)
    return avg.result()
  res['overlap_average_%dcrops_ms' % (len(ends_h) * len(ends_w))] = timeit(overlap, reps=3)
  print(json.dumps({'bench': 'ops around the hot path (ms per call, best of 5, 1 x MI355X)', 'result': res}))


if __name__ == '__main__':
  main()
