"""Wall time per training step of the mirrors, batch resident (median of 5 x 20 steps)."""
import os, sys, time, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
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
ts = []
for _ in range(5):
  t0 = time.perf_counter()
  for _ in range(20):
    util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
  ts.append((time.perf_counter() - t0) / 20 * 1e3)
print('train step, batch resident: median %.3f ms (%s) %s' % (statistics.median(ts), ' '.join('%.2f' % t for t in ts), ' '.join(sys.argv[1:])))
