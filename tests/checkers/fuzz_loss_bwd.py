"""Randomised check of the SegSortLoss kernels (forward and the backward tiles) against float64 autograd of the
reference formula (loss.py:15-82) on the device: random n / c / P / number of label sets / concentrations / modes /
groups / upstream-gradient scales, every backward route in turn (default = both contractions on the fp16 pipe,
HSGK_LOSS_BWD=mixed, HSGK_LOSS=fp32, HSGK_LOSS_BWD=generic).  Pixels whose 'segsort+' numerator cancels in fp32
(cond >= 20) get no upstream gradient.  Not part of the test suite; output appended to
profiles/r03_fuzz_parity.txt.

  python tests/checkers/fuzz_loss_bwd.py [n_cases] [seed]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hsg_amd.utils.segsort import loss as sl      # noqa: E402

ROUTES = (('h16', {}), ('mixed', {'HSGK_LOSS_BWD': 'mixed'}), ('fp32', {'HSGK_LOSS': 'fp32'}),
          ('generic', {'HSGK_LOSS_BWD': 'generic'}))


def ref_nll(e, sem, inst, p, psem, kappa, plus, qg=None, pg=None):
  sim = torch.exp(torch.mm(e, p.t()) * kappa)
  if qg is not None:
    sim = sim * (qg.view(-1, 1) == pg.view(1, -1)).double()
  own = sim.gather(1, inst.view(-1, 1))
  same = (sem.view(-1, 1) == psem.view(1, -1)).double()
  num = own
  cond = torch.ones_like(own)
  if plus:
    ss = (sim * same).sum(1, keepdim=True)
    sw = ss - own
    num = torch.where(sw > 0, sw, own)
    cond = (ss + own) / sw.abs().clamp_min(1e-300)
  den = (sim * (1.0 - same)).sum(1, keepdim=True) + num
  return -(num / den).log().view(-1), cond.view(-1)


def main():
  n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
  dev = torch.device('cuda:0')
  t0 = time.time()
  worst = {r: 0.0 for r, _ in ROUTES}
  worst_f = 0.0
  bad = 0
  for case in range(n_cases):
    c = int(rng.choice([64, 128, 256, 256, 256, 32, 96, 130, 384]))
    n = int(rng.integers(1, 6000))
    P = int(rng.integers(2, 1200))
    L = int(rng.integers(1, 4))
    grouped = L == 1 and bool(rng.integers(0, 2))
    g = torch.Generator(device=dev).manual_seed(int(rng.integers(1, 1 << 30)))
    pr = torch.nn.functional.normalize(torch.randn((P, c), device=dev, generator=g), dim=1)
    inst = torch.randint(0, P, (n,), device=dev, generator=g)
    e = torch.nn.functional.normalize(pr[inst] + float(rng.uniform(0.2, 1.0)) * torch.randn((n, c), device=dev, generator=g), dim=1)
    sets = []
    for l in range(L):
      classes = int(rng.integers(2, max(3, P // 2)))
      psem = torch.randint(0, classes, (P,), device=dev, generator=g)
      sem = psem[inst].clone()
      flip = torch.rand((n,), device=dev, generator=g) < 0.15
      sem[flip] = (sem[flip] + 1) % classes
      sets.append((sem, psem, float(rng.choice([4.0, 8.0, 10.0, 16.0])), str(rng.choice(['segsort+', 'segsort']))))
    qg = pg = None
    if grouped:
      pg = torch.sort(torch.randint(0, 3, (P,), device=dev, generator=g)).values
      qg = pg[inst]
    up = torch.exp(3.0 * np.log(10.0) * (2 * torch.rand((n,), device=dev, generator=g) - 1)).float()
    up[torch.rand((n,), device=dev, generator=g) < 0.1] = 0.0
    e2, p2 = e.double().requires_grad_(True), pr.double().requires_grad_(True)
    refs = []
    for sem, psem, k, m in sets:
      r, cond = ref_nll(e2, sem, inst, p2, psem, k, m == 'segsort+', qg, pg)
      up[cond >= 20.0] = 0.0
      refs.append((r, cond))
    wts = (1.0, 0.5, 2.0)
    sum((r * up.double()).sum() * w for (r, _), w in zip(refs, wts)).backward()
    line = 'case %4d: n=%5d c=%3d P=%4d L=%d %s' % (case, n, c, P, L, 'groups' if grouped else '      ')
    for route, env in ROUTES:
      for k_, v_ in (('HSGK_LOSS_BWD', ''), ('HSGK_LOSS', '')):
        os.environ.pop(k_, None)
      os.environ.update(env)
      et, pt = e.clone().requires_grad_(True), pr.clone().requires_grad_(True)
      if grouped:
        sem, psem, k, m = sets[0]
        nll = [sl.segsort_nll(et, sem, inst, pt, psem, k, m, pixel_groups=qg, prototype_groups=pg)]
      else:
        nll = [x.view(-1) for x in sl.segsort_losses(et, inst, pt, sets, reduction='none')]
      sum((x * up).sum() * w for x, w in zip(nll, wts)).backward()
      err = 0.0
      for got, ref in ((et.grad, e2.grad), (pt.grad, p2.grad)):
        err = max(err, (got.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30))
      worst[route] = max(worst[route], err)
      if route == 'h16':
        for x, (r, cond) in zip(nll, refs):
          live = torch.isfinite(r)
          fe = ((x.double() - r).abs() / cond.clamp_min(10.0))[live].max().item() if bool(live.any()) else 0.0
          worst_f = max(worst_f, fe)
      if not err <= 3e-5:
        bad += 1
        line += '  %s %.2e !!' % (route, err)
      else:
        line += '  %s %.1e' % (route, err)
    for k_ in ('HSGK_LOSS_BWD', 'HSGK_LOSS'):
      os.environ.pop(k_, None)
    if case % 20 == 19 or case == n_cases - 1:
      print(line)
  print('fuzz_loss_bwd: %d cases, %d route results above 3e-5 of the gradient scale; worst per route %s; forward nll '
        'worst %.1e per unit of max(10, cond) (%d s)' % (n_cases, bad, {k: '%.1e' % v for k, v in worst.items()}, worst_f,
                                                          time.time() - t0))
  return 1 if bad else 0


if __name__ == '__main__':
  sys.exit(main())
