"""How unbalanced is the M-step's wave ownership (label hash) on the bench input, and what would a
greedy size-balanced assignment give?  max over the 4 waves of the rows owned per 2048-row chunk,
relative to the perfect 512."""
import sys, torch
sys.path.insert(0, '.')
import hsg_amd.utils.segsort.common as sc
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(11)
x = torch.randn((2, 256, 448, 448), device=dev, generator=g)
for iters in (1, 5, 10):
  out = sc.segment_by_kmeans(x, None, [8, 8], iterations=iters)
  lab = (out[3][:448 * 448] % 64).cpu()          # image 0, cluster id within the image
  n = lab.numel()
  hash_max, lpt_max, idx_max, nch = 0.0, 0.0, 0.0, 0
  for c0 in range(0, n, 2048):
    l = lab[c0:c0 + 2048]
    cnt = torch.bincount(l, minlength=64)
    own = (torch.arange(64) ^ (torch.arange(64) >> 2) ^ (torch.arange(64) >> 4) ^ (torch.arange(64) >> 6)) & 3
    loads = torch.zeros(4, dtype=torch.long).scatter_add_(0, own, cnt)
    hash_max += loads.max().item()
    order = torch.argsort(cnt, descending=True)
    lp = [0, 0, 0, 0]
    for k in order.tolist():
      i = lp.index(min(lp)); lp[i] += int(cnt[k])
    lpt_max += max(lp)
    li = [0, 0, 0, 0]
    for k in range(64):                      # greedy in label order (no sort)
      if int(cnt[k]):
        i = li.index(min(li)); li[i] += int(cnt[k])
    idx_max += max(li)
    nch += 1
  print('iterations %2d: hash %.3f  greedy-by-size %.3f  greedy-in-label-order %.3f  (1.0 = perfectly balanced)'
        % (iters, hash_max / (n / 4), lpt_max / (n / 4), idx_max / (n / 4)))
