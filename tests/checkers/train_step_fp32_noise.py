"""Build container only (imports the real reference from /root/reference): how much of the
whole-train-step gradient is summation-order noise of the reference itself?  Re-runs the f14 step with
the similarity GEMM and the row sums of hsg/utils/segsort/loss.py:47-80 accumulated in float64 and
compares the embedding gradient with the float32 golden vector (tests/golden/f14_train_step_full.npz).
Measured: p50 4.4e-8, p90 7.1e-7, p99 7.2e-6, max 4.8e-5 on a gradient scale of 3.2e-2 (the
`same-label sum - own similarity` cancellation of 'segsort+'); hrchy_group_loss moves by 4e-5."""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/reference')
import inspect
import hsg.utils.segsort.common as ref_common
import hsg.utils.segsort.loss as ref_loss
import torch.nn.parallel.scatter_gather as sg
import hsg.models.utils as ref_mu
import hsg.models.embeddings.resnet_fcn_hsg as ref_model
import hsg.models.predictions.hsg as ref_pred
from tests import util as tutil
_src = inspect.getsource(ref_common.segment_by_kmeans).replace('cur_cluster_indices.device.index','(cur_cluster_indices.device.index or 0)')
_ns = dict(ref_common.__dict__); exec(_src, _ns)
ref_common.segment_by_kmeans = _ns['segment_by_kmeans']
sg.gather = lambda xs, dev=None, dim=0: torch.cat(list(xs), 0)
ref_mu.scatter_gather.gather = sg.gather
g = np.load('/root/repo/tests/golden/f14_train_step_full.npz')
inp = tutil.train_step_inputs(int(g['seed']))
loc_fn = lambda hw, dev: ref_common.generate_location_features(hw, dev, 'float') - 0.5
# variant: the similarity matrix of the loss computed in float64 and rounded back to fp32 --
# i.e. the reference formula with a different (more accurate) summation order of E.P^T only
orig_mm = torch.mm
def mm64(a, b):
    return orig_mm(a.double(), b.double()).float()
src = inspect.getsource(ref_loss._calculate_log_likelihood)
ns = dict(ref_loss.__dict__); ns['torch'] = type('T', (), {})()
import types
fake = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith('__')})
fake.mm = mm64
orig_sum = torch.sum
fake.sum = lambda t, *a, **k: orig_sum(t.double(), *a, **k).float()
ns['torch'] = fake
exec(src, ns)
ref_loss._calculate_log_likelihood = ns['_calculate_log_likelihood']
out = tutil.run_train_step(dict(embedding_cls=ref_model.MultiviewResnetFcn, prediction_cls=ref_pred.Hsg, model_utils=ref_mu, loc_fn=loc_fn), inp, 'cpu')
for k in ('img_sim_loss','hrchy_group_loss','clustering_loss'):
    print(k, float(out[k]), float(g[k]))
for k in ('grad',):
    err = np.abs(out[k].numpy() - g[k]); scale=np.abs(g[k]).max()
    print(k, 'reference(fp32 mm) vs reference(fp64 mm): scale %.3e p50 %.3e p90 %.3e p99 %.3e max %.3e' % (scale, np.quantile(err,0.5), np.quantile(err,0.9), np.quantile(err,0.99), err.max()))
