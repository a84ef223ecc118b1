"""torch-CPU restatement of the reference operator sequence (TEST / BASELINE
infrastructure only; nothing under hsg_amd/ imports it).

This is the `cpu_baseline` of bench.py: the same ATen op sequence the
reference executes on CPU for hsg/utils/segsort/common.py:270-408 (MKL sgemm,
scatter_add_, sorted unique), so its wall time on the GPU box's host cores is
what "the reference's CPU path" costs there.  Its outputs are checked against
the golden vectors in tests/test_oracle_golden.py.  The bit-exact checker of
the HIP kernels is oracle/hsg_oracle.c, not this file (ATen leaves summation
order unspecified).
"""
import torch

EPS = 1e-12


def l2_normalize(t, eps=EPS):
  # general/common.py:116-120
  nrm = torch.norm(t, dim=-1, keepdim=True)
  nrm = torch.where(nrm >= eps, nrm, torch.full_like(nrm, eps))
  return t / nrm


def mean_directions(rows, assign, count):
  # segsort/common.py:34-39 (M-step)
  acc = torch.zeros((count, rows.shape[-1]), dtype=rows.dtype)
  acc.scatter_add_(0, assign.view(-1, 1).expand(-1, rows.shape[-1]), rows)
  return l2_normalize(acc)


def nearest(rows, protos):
  # segsort/common.py:62-64 (E-step)
  return torch.argmax(torch.mm(rows, protos.t()), 1)


def lloyd(rows, assign, count, iterations):
  # segsort/common.py:90-95
  for _ in range(iterations):
    assign = nearest(rows, mean_directions(rows, assign, count))
  return assign


def grid_seeds(grid, hw):
  # segsort/common.py:145-151
  ys = torch.linspace(0, grid[0] - 1, hw[0]).round_().long()
  xs = torch.linspace(0, grid[1] - 1, hw[1]).round_().long()
  return ys.view(-1, 1) + (ys.max() + 1) * xs.view(1, -1)


def location_map(hw):
  # segsort/common.py:175-187 + :316
  gy, gx = torch.meshgrid(torch.linspace(0, 1, hw[0]), torch.linspace(0, 1, hw[1]),
                          indexing='ij')
  return torch.stack([gy, gx], 2) - 0.5


def segment_by_kmeans(nchw, labels=None, grid=(5, 5), loc=None, ignore_index=None,
                      iterations=10, gpu_id=0):
  """Same outputs as segsort/common.py:270-408 for CPU tensors."""
  feats = l2_normalize(nchw.permute(0, 2, 3, 1).contiguous())
  B, H, W, C = feats.shape
  if loc is None:
    loc = location_map((H, W))
  if loc.dim() == 3:
    loc = loc.view(1, H, W, 2).expand(B, H, W, 2)
  seeds = grid_seeds(grid, (H, W)).view(1, H, W).expand(B, H, W)
  if labels is None:
    labels = torch.zeros((B, H, W), dtype=torch.long)
  keep = {k: [] for k in ('lab', 'clu', 'bat', 'emb', 'eloc')}
  for b in range(B):
    lab = labels[b].reshape(-1)
    clu = torch.unique(seeds[b].reshape(-1), return_inverse=True)[1]
    count = clu.max() + 1
    emb = feats[b].view(-1, C)
    eloc = l2_normalize(torch.cat([emb, loc[b].reshape(-1, 2)], -1))
    if ignore_index is not None:
      sel = (lab != ignore_index).nonzero().view(-1)
      lab, clu = lab.index_select(0, sel), clu.index_select(0, sel)
      emb, eloc = emb.index_select(0, sel), eloc.index_select(0, sel)
    if emb.shape[0] > 0:
      clu = lloyd(eloc, clu, count, iterations)
    keep['lab'].append(lab)
    keep['clu'].append(clu)
    keep['bat'].append(torch.full_like(clu, b + B * gpu_id))
    keep['emb'].append(emb)
    keep['eloc'].append(eloc)
  lab, clu, bat = (torch.cat(keep[k], 0) for k in ('lab', 'clu', 'bat'))
  emb, eloc = torch.cat(keep['emb'], 0), torch.cat(keep['eloc'], 0)
  clu = torch.unique(bat * (clu.max() + 1) + clu, return_inverse=True)[1]   # :398-401
  clu = torch.unique(lab + clu * (lab.max() + 1), return_inverse=True)[1]   # :404 / :212-214
  return emb, eloc, lab, clu, bat
