#!/usr/bin/env python3
"""Micro-benchmark of the Lloyd half-steps (E = assign, M = accumulate +
finalize) through the C ABI, on synthetic unit rows.  Used for kernel tuning
and as the target command of rocprofv3 runs.

  python tools/bench_kernels.py [--B 48 --HW 200704 --D 258 --K 64 --reps 5]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--B', type=int, default=48)
  ap.add_argument('--HW', type=int, default=448 * 448)
  ap.add_argument('--D', type=int, default=258)
  ap.add_argument('--K', type=int, default=64)
  ap.add_argument('--reps', type=int, default=5)
  ap.add_argument('--only', default='em')
  ap.add_argument('--lloyd-warm', type=int, default=3)
  ap.add_argument('--unit', type=int, default=1, help='1 = bf16 split fast path, 0 = fp32 kernel')
  a = ap.parse_args()
  import torch
  from hsg_amd import _lib
  L = _lib.lib()
  dev = torch.device('cuda:0')
  n = a.B * a.HW
  g = torch.Generator(device=dev)
  g.manual_seed(1)
  x = torch.randn((n, a.D), device=dev, generator=g)
  x = x / x.norm(dim=1, keepdim=True)
  lab = torch.randint(0, a.K, (n,), device=dev, dtype=torch.int32, generator=g)
  cent = torch.empty((a.B, a.K, a.D), device=dev)
  out = torch.empty((n,), device=dev, dtype=torch.int32)
  wsb = L.hsgk_lloyd_workspace_bytes(a.B, a.HW, a.D, a.K)
  ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
  st = _lib.stream_ptr()

  def m():
    _lib.check(L.hsgk_lloyd_mstep(x.data_ptr(), a.B, a.HW, a.D, a.K, lab.data_ptr(),
                                  cent.data_ptr(), ws.data_ptr(), wsb, st))

  def e():
    _lib.check(L.hsgk_lloyd_estep(x.data_ptr(), a.B, a.HW, a.D, a.K, cent.data_ptr(),
                                  out.data_ptr(), a.unit, ws.data_ptr(), wsb, st))

  # a few real Lloyd iterations first so that labels / centroids look like the
  # operator's steady state (random labels give 64 near-identical centroids)
  def relabel():
    lab.copy_(out)
  for _ in range(a.lloyd_warm):
    m(); e(); relabel()
  m(); e(); torch.cuda.synchronize()
  _lib.profile_enable(True)
  _lib.profile_collect()
  for _ in range(a.reps):
    if 'm' in a.only:
      m()
    if 'e' in a.only:
      e()
  torch.cuda.synchronize()
  prof = _lib.profile_collect()
  res = {}
  gb = n * (4 * a.D + 8) / 1e9
  for k, (ms, cnt) in prof.items():
    if cnt:
      res[k] = {'ms': round(ms / cnt, 4), 'GB/s(4D+8)': round(gb / (ms / cnt) * 1e3, 1)}
  if 'assign' in res:
    res['assign']['TFLOP/s'] = round(2.0 * a.D * a.K * n / (res['assign']['ms'] * 1e-3) / 1e12, 1)
  if a.unit:
    q = torch.zeros((2,), dtype=torch.int64, device=dev)
    _lib.check(L.hsgk_lloyd_requeued_rows(a.B, a.HW, a.D, a.K, ws.data_ptr(), wsb, q.data_ptr(), st))
    res['requeued_fraction'] = round(q[0].item() / n, 5)
    res['fp16_undecided_fraction'] = round(q[1].item() / n, 5)
  print(json.dumps({'shape': vars(a), 'result': res}))


if __name__ == '__main__':
  main()
