"""CPU-side op counts of one training step of the mirrors, per phase (torch profiler)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import util
from hsg_amd.models import utils as mu
from hsg_amd.models.embeddings import resnet_fcn_hsg as emb_mod
from hsg_amd.models.embeddings import hierarchy as hz
from hsg_amd.models.predictions import hsg as pred_mod
from hsg_amd.utils.segsort import common as sc
from torch.profiler import profile, ProfilerActivity, record_function
util.TRAIN_STEP.update(B=4, C=128, H=56, W=56, grid=(4, 4), iters=15, M=256, KF=8, KC=4, label_divisor=255,
                       ignore=255, kappa=16.0, dmon_knn=4, image_ids=[0, 1, 0, 1])
dev = torch.device('cuda:0')
inp = util.train_step_inputs(1234)
emb_cls = [getattr(emb_mod, n) for n in dir(emb_mod) if n.startswith('Multiview')][0]
mods = dict(embedding_cls=emb_cls, prediction_cls=pred_mod.Hsg, model_utils=mu,
            loc_fn=lambda hw, d: sc.generate_location_features(hw, d, 'float') - 0.5)
def wrap(mod, name, tag):
  fn = getattr(mod, name)
  def f(*a, **k):
    with record_function('PHASE_' + tag):
      return fn(*a, **k)
  setattr(mod, name, f)
for name, tag in (('calculate_kmeans_prototypes', 'kmeans_prototypes'), ('hierarchical_grouping_from_logits', 'hier_grouping'),
                  ('collect_nd_coarser_prototype', 'collect_nd'), ('collect_pixel_hierarchical_clustering_indices', 'collect_pixel'),
                  ('transformer_clustering_tail', 'tc_tail')):
  if hasattr(hz, name):
    wrap(hz, name, tag)
wrap(sc, 'segment_by_kmeans', 'segment_by_kmeans')
wrap(emb_cls, 'generate_clusters', 'generate_clusters')
wrap(pred_mod.Hsg, 'losses', 'hsg_losses')
for _ in range(3):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
  util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
evs = prof.events()
phases = [e for e in evs if e.name.startswith('PHASE_')]
for p in phases:
  inside = [k for k in evs if k.time_range.start >= p.time_range.start and k.time_range.end <= p.time_range.end and k.name.startswith('aten::') and k.cpu_parent is not None and not (k.cpu_parent.name.startswith('aten::'))]
  c = collections.Counter(k.name for k in inside)
  print('%-22s %7.2f ms  %4d top-level aten ops: %s' % (p.name[6:], (p.time_range.end - p.time_range.start) / 1e3, len(inside), dict(c.most_common(8))))
