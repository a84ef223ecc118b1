"""cProfile of the Python side of the training step of the mirrors (batch resident): functions by own time."""
import os, sys, cProfile, pstats, io
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
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats('tottime')
st.print_stats(45)
txt = out.getvalue().replace(ROOT + '/', '')
print('20 steps; times below are totals over them')
print('\n'.join(l[:150] for l in txt.splitlines()[4:]))
