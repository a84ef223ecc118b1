"""SegSortLoss forward / backward at (N, C, P) for rocprofv3 --kernel-trace --stats: python tools/probes/loss_prof.py N C P [engine]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hsg_amd.utils.segsort.loss import SegSortLoss
n, c, P = (int(v) for v in sys.argv[1:4])
if len(sys.argv) > 4:
  os.environ['HSGK_LOSS'] = sys.argv[4]
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
pr = torch.nn.functional.normalize(torch.randn((P, c), device=dev, generator=g), dim=1)
ins = torch.randint(0, P, (n,), device=dev, generator=g)
e = torch.nn.functional.normalize(pr[ins] + 0.35 * torch.randn((n, c), device=dev, generator=g), dim=1)
ps = torch.arange(P, device=dev) % 21
se = ps[ins]
for _ in range(5):
  a = e.detach().requires_grad_(True)
  b = pr.detach().requires_grad_(True)
  SegSortLoss(16, 'segsort+')(a, se, ins, b, ps).backward()
torch.cuda.synchronize()
