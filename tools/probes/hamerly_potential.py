"""Would Hamerly-style bounds let the E-step skip rows on the bench input?  For iterations it -> it+1:
centroid movement delta_k = |c_k' - c_k| and the fraction of rows whose exact margin (best - second
best score under the OLD centroids) exceeds delta_best + max_other delta (those rows provably keep
their label)."""
import sys, torch
sys.path.insert(0, '.')
import hsg_amd.utils.segsort.common as sc
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((1, 256, 448, 448), device=dev, generator=g)
outs = {}
for it in (0, 1, 2, 4, 6, 8, 9, 10):
  o = sc.segment_by_kmeans(x, None, [8, 8], iterations=it) if it > 0 else None
  outs[it] = o
emb_loc = outs[1][1]
def cents(labels):
  return sc.calculate_prototypes_from_labels(emb_loc, labels, 64)
for a, b in ((1, 2), (4, 5), (8, 9), (9, 10)):
  la = sc.segment_by_kmeans(x, None, [8, 8], iterations=a)[3] % 64
  lb = sc.segment_by_kmeans(x, None, [8, 8], iterations=b)[3] % 64
  ca, cb = cents(la), cents(lb)          # centroids used by E-step a+1 and b+1 (within fp32 order)
  delta = (cb - ca).norm(dim=1)
  s = emb_loc @ ca.t()
  top2 = s.topk(2, dim=1)
  margin = top2.values[:, 0] - top2.values[:, 1]
  best = top2.indices[:, 0]
  dmax_other = torch.where(torch.arange(64, device=dev)[None, :] == best[:, None], torch.zeros((), device=dev),
                           delta[None, :].expand(best.numel(), 64)).max(dim=1).values
  need = delta[best] + dmax_other
  print('labels after it %d -> %d: delta mean %.4f max %.4f | margin median %.4f | rows provably unchanged %.3f | labels changed %.4f'
        % (a, b, delta.mean().item(), delta.max().item(), margin.median().item(), (margin > need).float().mean().item(),
           (la != lb).float().mean().item()))
