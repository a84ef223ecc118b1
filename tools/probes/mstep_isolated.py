"""The exact-sum update kernel alone at cfg2 size through hsgk_lloyd_mstep_exact: random unit rows, int32 labels,
`frac` of the rows changed between labels_prev and labels -- is the kernel slow by itself (vs tools/probes/gather_steps:
0.55 ms at 28.6 %), or only inside segment_by_kmeans?"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
L = _lib.lib()
dev = torch.device('cuda:0')
B, R, d, K = 48, 448 * 448, 258, 64
n = B * R
x = torch.randn((n, d), device=dev)
x /= x.norm(dim=1, keepdim=True)
g = torch.Generator(device=dev).manual_seed(1)
prev = torch.randint(0, K, (n,), device=dev, dtype=torch.int32, generator=g)
sums = torch.zeros((B, K, d), dtype=torch.int64, device=dev)
cent = torch.empty((B, K, d), device=dev)
wsb = L.hsgk_lloyd_workspace_bytes(B, R, d, K)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
st = _lib.stream_ptr()
_lib.check(L.hsgk_lloyd_mstep_exact(x.data_ptr(), B, R, d, K, None, prev.data_ptr(), sums.data_ptr(), cent.data_ptr(), ws.data_ptr(), wsb, st))
torch.cuda.synchronize()
for frac in (0.286, 0.12, 0.026):
  ch = torch.rand((n,), device=dev, generator=g) < frac
  cur = torch.where(ch, (prev + 1 + torch.randint(0, K - 1, (n,), device=dev, dtype=torch.int32, generator=g)) % K, prev).to(torch.int32)
  ts = []
  a, b = prev, cur
  for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.hsgk_lloyd_mstep_exact(x.data_ptr(), B, R, d, K, a.data_ptr(), b.data_ptr(), sums.data_ptr(), cent.data_ptr(), ws.data_ptr(), wsb, st))
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
    a, b = b, a                      # back and forth: the same rows change every time
  print('changed %.1f %%: whole call (setup + update + finalize) %s ms' % (100 * frac, ' '.join('%.3f' % t for t in ts)))

# ---- the same call on the REAL first update of a cfg2 call: rows = the operator's embeddings_with_loc, labels_prev = the
# seed grid, labels = the labels after one iteration
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
del x, prev, cur, ch
torch.cuda.empty_cache()
xin = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (48, 256, 448, 448), 'iid', dev)
out = sc.segment_by_kmeans(xin, None, [8, 8], iterations=1)
rows = out[1].contiguous()
cur = (out[3] - 64 * out[4]).to(torch.int32).contiguous()
seeds = sc.initialize_cluster_labels([8, 8], [448, 448], dev).view(-1)
seeds = torch.unique(seeds, return_inverse=True)[1].to(torch.int32)
prev = seeds.repeat(48).contiguous()
del xin, out
print('real first update: %.1f %% of the rows change' % (100.0 * (prev != cur).float().mean().item()))
_lib.check(L.hsgk_lloyd_mstep_exact(rows.data_ptr(), B, R, d, K, None, prev.data_ptr(), sums.data_ptr(), cent.data_ptr(), ws.data_ptr(), wsb, st))
torch.cuda.synchronize()
ts = []
a, b = prev, cur
for rep in range(6):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  _lib.check(L.hsgk_lloyd_mstep_exact(rows.data_ptr(), B, R, d, K, a.data_ptr(), b.data_ptr(), sums.data_ptr(), cent.data_ptr(), ws.data_ptr(), wsb, st))
  e1.record(); torch.cuda.synchronize()
  ts.append(e0.elapsed_time(e1))
  a, b = b, a
print('real labels: whole call %s ms' % ' '.join('%.3f' % t for t in ts))
# how the changed rows are spread: per 2048-row chunk and per 37 K-row workgroup range
chg = (prev != cur).view(-1, 2048).float().sum(1)
print('changed rows per chunk: mean %.0f  min %.0f  max %.0f' % (chg.mean().item(), chg.min().item(), chg.max().item()))
nc = chg.shape[0]
per_wg = torch.stack([chg[(i * nc) // 256:((i + 1) * nc) // 256].sum() for i in range(256)])
print('changed rows per workgroup range: mean %.0f  min %.0f  max %.0f' % (per_wg.mean().item(), per_wg.min().item(), per_wg.max().item()))
