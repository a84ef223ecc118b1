"""Top-level aten ops of one training step of the mirrors attributed to the Python line that issued them."""
import os, sys, collections
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
for _ in range(3):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
import time, traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
class Log(TorchDispatchMode):
  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    t0 = time.perf_counter()
    out = func(*args, **(kwargs or {}))
    dt = time.perf_counter() - t0
    where = 'autograd/backward'
    for fr in reversed(traceback.extract_stack()[:-1]):
      if fr.filename.startswith(ROOT) and 'tools/probes' not in fr.filename:
        where = '%s:%d' % (fr.filename.replace(ROOT + '/', ''), fr.lineno); break
    a = agg[where]; a[0] += 1; a[1] += dt * 1e3; a[2][str(func).replace('aten.', '')] += 1
    return out
with Log():
  util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
tot = sum(v[1] for v in agg.values())
print('total dispatched op time %.2f ms over %d ops' % (tot, sum(v[0] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
  print('%6.3f ms %3d  %-58s %s' % (v[1], v[0], k[:58], dict(v[2].most_common(5))))
