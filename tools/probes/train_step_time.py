"""Wall time of one whole training step of the mirrors (everything after the backbone: clustering, hierarchy,
three prototype exchanges, Hsg.losses forward + backward) at the reference's training hyper-parameters, with the
time of its parts (torch profiler: GPU time per kernel family, CPU wall)."""
import os, sys, time, types
import numpy as np
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
inp = util.train_step_inputs(1234)
emb_cls = [getattr(emb_mod, n) for n in dir(emb_mod) if n.startswith('Multiview')][0]
mods = dict(embedding_cls=emb_cls, prediction_cls=pred_mod.Hsg, model_utils=mu,
            loc_fn=lambda hw, d: sc.generate_location_features(hw, d, 'float') - 0.5)
for _ in range(3):
  out = util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
  out = util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
print('whole step (incl. host->device copies of the inputs): %.2f ms' % ((time.perf_counter() - t0) / n * 1e3))
inp_dev = util.device_inputs(inp, dev)
for _ in range(3):
  util.run_train_step(mods, inp_dev, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
  util.run_train_step(mods, inp_dev, dev)
torch.cuda.synchronize()
print('whole step, batch resident on the device: %.2f ms' % ((time.perf_counter() - t0) / n * 1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
  for _ in range(3):
    out = util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(e.device_time_total for e in ka if e.device_time_total) / 3e3
print('GPU kernel time per step: %.2f ms in %d launches' % (tot, sum(e.count for e in ka if e.device_time_total) // 3))
rows = sorted([e for e in ka if e.device_time_total], key=lambda e: -e.device_time_total)[:14]
for e in rows:
  print('  %-70s %4d x  %8.1f us total per step' % (e.key[:70], e.count // 3, e.device_time_total / 3.0))

# ---- launches and GPU time per phase of the step (record_function ranges around the calls of run_train_step)
import contextlib
from torch.profiler import record_function
orig = dict(gen=emb_cls.generate_clusters, exch=mu.gather_clustering_and_update_prototypes,
            maps=mu.gather_and_update_cluster_mappings, fwd=pred_mod.Hsg.forward)
def wrap(name, fn):
  def f(*a, **k):
    with record_function('PHASE_' + name):
      return fn(*a, **k)
  return f
emb_cls.generate_clusters = wrap('generate_clusters', orig['gen'])
mu.gather_clustering_and_update_prototypes = wrap('exchange', orig['exch'])
mu.gather_and_update_cluster_mappings = wrap('mappings', orig['maps'])
pred_mod.Hsg.forward = wrap('losses_forward', orig['fwd'])
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
  out = util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
evs = prof.events()
phases = [e for e in evs if e.name.startswith('PHASE_')]
kern = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA]
print('phases of one step (CPU wall of the call, kernels launched inside it):')
for p in phases:
  inside = [k for k in evs if k.device_type == torch.autograd.DeviceType.CPU and k.time_range.start >= p.time_range.start and k.time_range.end <= p.time_range.end and k.name.startswith(('hipLaunchKernel', 'hipExtModuleLaunchKernel', 'hipModuleLaunchKernel', 'hipMemcpy', 'hipMemset'))]
  print('  %-22s %8.2f ms CPU, %4d launches / copies' % (p.name[6:], (p.time_range.end - p.time_range.start) / 1e3, len(inside)))
