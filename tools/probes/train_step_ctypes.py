"""Host time of every libhsgk entry point inside the training step (batch resident): total and per call over 20 steps.
A call that takes far longer than the ~10 us of a kernel launch is waiting for something (a full queue, a
synchronising runtime call)."""
import os, sys, time, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
from hsg_amd import _lib
from hsg_amd.models import utils as mu
from hsg_amd.models.embeddings import resnet_fcn_hsg as emb_mod
from hsg_amd.models.predictions import hsg as pred_mod
from hsg_amd.utils.segsort import common as sc
util.TRAIN_STEP.update(B=4, C=128, H=56, W=56, grid=(4, 4), iters=15, M=256, KF=8, KC=4, label_divisor=255,
                       ignore=255, kappa=16.0, dmon_knn=4, image_ids=[0, 1, 0, 1])
dev = torch.device('cuda:0')
inp = util.device_inputs(util.train_step_inputs(1234), dev)
emb_cls = [getattr(emb_mod, n) for n in dir(emb_mod) if n.startswith('Multiview')][0]
mods = dict(embedding_cls=emb_cls, prediction_cls=pred_mod.Hsg, model_utils=mu,
            loc_fn=lambda hw, d: sc.generate_location_features(hw, d, 'float') - 0.5)
for _ in range(10):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
real = _lib.lib()
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])


class Proxy:
  def __getattr__(self, name):
    fn = getattr(real, name)

    def timed(*a):
      t0 = time.perf_counter()
      r = fn(*a)
      dt = time.perf_counter() - t0
      e = acc[name]
      e[0] += dt; e[1] += 1; e[2] = max(e[2], dt)
      return r
    return timed


_lib._lib = Proxy()
t0 = time.perf_counter()
for _ in range(20):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
_lib._lib = real
tot = sum(v[0] for v in acc.values()) / 20 * 1e3
print('wall per step %.3f ms (with the timers); inside libhsgk entry points %.3f ms per step, %d calls per step'
      % (wall, tot, sum(v[1] for v in acc.values()) // 20))
for name, (t, n, mx) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:25]:
  print('  %-44s %7.1f us per step  %3d calls per step  %6.1f us per call  max %7.1f us' % (name, t / 20 * 1e6, n // 20, t / n * 1e6, mx * 1e6))
