import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from hsg_amd.utils.segsort import common as sc
x = torch.randn((48, 256, 28, 28), device='cuda:0')
for _ in range(12):
  out = sc.segment_by_kmeans(x, None, [8, 8], iterations=10)
torch.cuda.synchronize()
