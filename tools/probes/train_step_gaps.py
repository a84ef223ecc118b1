"""Idle gaps of the device inside one training step of the mirrors (batch resident): for every gap of more than
20 us between consecutive device operations, the operation that ended it and the CPU-side op that launched it."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
from hsg_amd.models import utils as mu
from hsg_amd.models.embeddings import resnet_fcn_hsg as emb_mod
from hsg_amd.models.predictions import hsg as pred_mod
from hsg_amd.utils.segsort import common as sc
from torch.profiler import profile, ProfilerActivity
util.TRAIN_STEP.update(B=4, C=128, H=56, W=56, grid=(4, 4), iters=15, M=256, KF=8, KC=4, label_divisor=255,
                       ignore=255, kappa=16.0, dmon_knn=4, image_ids=[0, 1, 0, 1])
dev = torch.device('cuda:0')
inp = util.device_inputs(util.train_step_inputs(1234), dev)
emb_cls = [getattr(emb_mod, n) for n in dir(emb_mod) if n.startswith('Multiview')][0]
mods = dict(embedding_cls=emb_cls, prediction_cls=pred_mod.Hsg, model_utils=mu,
            loc_fn=lambda hw, d: sc.generate_location_features(hw, d, 'float') - 0.5)
for _ in range(5):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
print('wall per step, batch resident: %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
  util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
evs = prof.events()
ks = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in ks) / 1e3
span = (ks[-1].time_range.end - ks[0].time_range.start) / 1e3
print('device: %d operations, busy %.3f ms inside a span of %.3f ms' % (len(ks), busy, span))
cpu = sorted([e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and not e.name.startswith(('hip', 'PHASE'))],
             key=lambda e: e.time_range.start)
syncs = [e for e in evs if e.name in ('hipStreamSynchronize', 'hipDeviceSynchronize', 'hipMemcpyWithStream', 'hipMemcpyAsync', 'hipEventSynchronize')]
print('host waits: %d (%s)' % (len(syncs), ', '.join('%s x%d' % (n, sum(1 for s in syncs if s.name == n)) for n in sorted({s.name for s in syncs}))))
import bisect
cpu_start = [c.time_range.start for c in cpu]
gaps = []
for a, b in zip(ks[:-1], ks[1:]):
  g = (b.time_range.start - a.time_range.end) / 1e3
  if g > 0.02:
    i = bisect.bisect_right(cpu_start, b.time_range.start) - 1      # the last CPU op that started before b
    gaps.append((g, a.name[:40], b.name[:48], cpu[i].name[:40] if i >= 0 else '?'))
print('idle gaps > 20 us: %d, %.3f ms in total' % (len(gaps), sum(g[0] for g in gaps)))
for g in sorted(gaps, reverse=True)[:40]:
  print('  %7.3f ms  after %-40s before %-48s cpu: %s' % g)
