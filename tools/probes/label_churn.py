"""How many rows change label per Lloyd iteration on the bench's i.i.d. input (decides whether an
incremental M-step could pay).  Runs on the GPU box."""
import sys, torch
sys.path.insert(0, '.')
import hsg_amd.utils.segsort.common as sc
dev = torch.device('cuda:0')
gen = torch.Generator(device=dev); gen.manual_seed(1234)
B, C, H, W = 4, 256, 448, 448
x = torch.randn((B, C, H, W), device=dev, generator=gen)
prev = None
for it in range(1, 11):
  out = sc.segment_by_kmeans(x, None, [8, 8], iterations=it)
  ci = out[3].clone()          # cluster_indices (dense relabel; stable while no cluster empties)
  if prev is not None:
    ch = (ci != prev).float().mean().item()
    print('iteration %2d: %.4f of the rows changed label, %d segments' % (it, ch, int(ci.max().item()) + 1))
  prev = ci
