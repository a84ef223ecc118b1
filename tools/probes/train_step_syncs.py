"""Every host <-> device transfer of one training step of the mirrors (batch resident) with the Python line that
issued it: device -> host reads stall the launch queue, host -> device uploads of scalars cost a pageable copy each."""
import os, sys, collections, traceback
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
for _ in range(3):
  util.run_train_step(mods, inp, dev)
torch.cuda.synchronize()
from torch.utils._python_dispatch import TorchDispatchMode
log = collections.Counter()
nops = [0]
def devs(x):
  out = set()
  def walk(a):
    if torch.is_tensor(a): out.add(a.device.type)
    elif isinstance(a, (list, tuple)):
      for b in a: walk(b)
  walk(x)
  return out
class Log(TorchDispatchMode):
  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    nops[0] += 1
    out = func(*args, **(kwargs or {}))
    name = str(func).replace('aten.', '')
    din, dout = devs(args), devs(out)
    kind = None
    if name.startswith('_local_scalar_dense') and 'cuda' in din: kind = 'D2H read'
    elif 'cuda' in din and 'cpu' in dout: kind = 'D2H copy'
    elif 'cpu' in din and 'cuda' in dout: kind = 'H2D copy'
    elif name.startswith('copy_') and len(din) > 1: kind = 'cross copy_'
    elif name.startswith(('nonzero', '_unique', 'unique', 'masked_select')) and 'cuda' in din: kind = 'sync op (%s)' % name.split('.')[0]
    elif name.startswith(('index_fill', 'masked_fill', 'fill_', 'full', 'scalar_tensor')) and 'cuda' in (din | dout) and any(isinstance(a, (int, float)) for a in args): kind = None
    if kind:
      where = 'autograd/backward'
      for fr in reversed(traceback.extract_stack()[:-1]):
        if fr.filename.startswith(ROOT) and 'tools/probes' not in fr.filename:
          where = '%s:%d' % (fr.filename.replace(ROOT + '/', ''), fr.lineno); break
      log[(kind, where, name)] += 1
    return out
with Log():
  util.run_train_step(mods, inp, dev)
  torch.cuda.synchronize()
print('%d dispatched ops; host <-> device transfers:' % nops[0])
for (kind, where, name), n in sorted(log.items(), key=lambda kv: (kv[0][0], kv[0][1])):
  print('  %-22s %2d x  %-58s %s' % (kind, n, where, name))
