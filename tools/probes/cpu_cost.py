import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from hsg_amd.utils.segsort import common as sc
dev = torch.device('cuda:0')
x = torch.randn((4, 128, 56, 56), device=dev)
for _ in range(5): sc.segment_by_kmeans(x, None, [4, 4], iterations=15)
torch.cuda.synchronize()
# tiny GPU work: 1 iteration on a tiny map -> CPU cost per call dominates
xs = torch.randn((1, 128, 8, 8), device=dev)
for _ in range(5): sc.segment_by_kmeans(xs, None, [2, 2], iterations=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): sc.segment_by_kmeans(xs, None, [2, 2], iterations=1)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('tiny call: CPU enqueue %.1f us per call, total %.1f us per call' % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): sc.segment_by_kmeans(xs, None, [2, 2], iterations=1)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
