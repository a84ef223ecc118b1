import sys, time, torch
sys.path.insert(0, '.')
import hsg_amd.utils.segsort.common as sc
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.randn((4, 256, 768, 768), device=dev, generator=g)
for it in range(2):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  out = sc.segment_by_kmeans(x, None, [16, 16], iterations=10)
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('cfg4-like 4x256x768x768 K=256: %.1f ms, %.1f Mpx/s, segments %d' % (dt * 1e3, 4 * 768 * 768 / dt / 1e6, int(out[3].max()) + 1))
cidx = out[3]
# properties: every pixel's cluster is its nearest final centroid under the exact kernel? (idempotence of one more E-step)
emb_loc = out[1]
protos = sc.calculate_prototypes_from_labels(emb_loc[:768*768], cidx[:768*768])
near = sc.find_nearest_prototypes(emb_loc[:768*768], protos)
print('fixed-point fraction (image 0):', float((near == cidx[:768*768]).float().mean()))
