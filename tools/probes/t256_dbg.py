"""E-step of the tile-order K <= 256 kernel against the exact kernel on one 768 x 768 image; details of differing rows"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd import _lib
from hsg_amd.utils import synth
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
L = _lib.lib()
B, C, H, W, K = 1, 256, 768, 768, 256
x = synth.device_embeddings_nchw(synth.SEED_BASE + 4, (B, C, H, W), 'iid', dev)
emb, eloc, lab, cidx, bidx = sc.segment_by_kmeans(x, None, [16, 16], iterations=0)
D = C + 2
n = eloc.shape[0]
from hsg_amd import ops
cent = sc.calculate_prototypes_from_labels(eloc, cidx, K).contiguous().view(B, K, D)
wsb = L.hsgk_lloyd_workspace_bytes(B, H * W, D, K)
ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
got = {}
for unit in (2, 0):
  out = torch.full((n,), -1, dtype=torch.int32, device=dev)
  _lib.check(L.hsgk_lloyd_estep(eloc.data_ptr(), B, H * W, D, K, cent.data_ptr(), out.data_ptr(), unit, ws.data_ptr(), wsb, _lib.stream_ptr()))
  got[unit] = out.clone()
torch.cuda.synchronize()
bad = torch.nonzero(got[2] != got[0]).view(-1)
print('rows', n, 'differing', bad.numel())
c64 = cent[0].double()
ch = cent[0].half().double()
for r in bad[:12].tolist():
  xr = eloc[r].double()
  s = c64 @ xr
  sa = ch @ eloc[r].half().double()
  o = torch.argsort(s, descending=True)[:6]
  print('row', r, 'lane j', r % 32, 'got', int(got[2][r]), 'exact', int(got[0][r]))
  print('   exact top', [(int(k), float(s[k])) for k in o])
  oa = torch.argsort(sa, descending=True)[:6]
  print('   approx top', [(int(k), float(sa[k]), 'half', (int(k) % 32 // 4) % 2, 'pass', int(k) // 64) for k in oa])
