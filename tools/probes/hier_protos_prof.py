"""Device operations inside calculate_kmeans_prototypes at cfg4's per-GPU size (4 x 768 x 768 pixels, 256 segments per image)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hsg_amd.models.embeddings import hierarchy as hz
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
B, HW, C, K = 4, 768 * 768, 256, 256
g = torch.Generator(device=dev).manual_seed(3)
n = B * HW
emb = torch.nn.functional.normalize(torch.randn((n, C), device=dev, generator=g), dim=1)
pos = torch.randn((n, C), device=dev, generator=g)
bidx = torch.arange(B, device=dev).repeat_interleave(HW)
cidx = torch.randint(0, K, (n,), device=dev, generator=g) + bidx * K
lab = torch.zeros((n,), dtype=torch.long, device=dev)
f = lambda: hz.calculate_kmeans_prototypes(emb, cidx, bidx, pos, lab, None, label_divisor=2048, max_num_clusters=K)
for _ in range(3): f()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
  f(); torch.cuda.synchronize()
ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
c = collections.defaultdict(lambda: [0, 0.0])
for e in ks:
  c[e.name][0] += 1; c[e.name][1] += (e.time_range.end - e.time_range.start) / 1e3
print('%d device operations, %.3f ms' % (len(ks), sum(v[1] for v in c.values())))
for nme, (k, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:16]:
  print('%8.3f ms %3d  %s' % (t, k, nme[:100]))
