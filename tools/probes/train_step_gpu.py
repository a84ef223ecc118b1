"""Wall time per training step of the mirrors vs the sum of its device kernel durations (torch profiler)."""
import os, sys, time, collections
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
inp = util.device_inputs(util.train_step_inputs(1234), dev)      # the batch resident, as a training loop's loader leaves it
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
print('wall per step %.3f ms' % ((time.perf_counter() - t0) / 20 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
  util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
ks = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = sum(e.time_range.end - e.time_range.start for e in ks) / 1e3
print('device kernels: %d launches, %.3f ms total' % (len(ks), tot))
c = collections.defaultdict(lambda: [0, 0.0])
for e in ks:
  c[e.name][0] += 1; c[e.name][1] += (e.time_range.end - e.time_range.start) / 1e3
for n, (k, t) in sorted(c.items(), key=lambda kv: -kv[1][1])[:60]:
  print('%8.3f ms %4d  %s' % (t, k, n[:110]))
