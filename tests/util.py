"""Shared helpers: rebuild fixture inputs from the portable generator."""
import os

import numpy as np

from hsg_amd.utils import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROW_STRIDE = 29        # tools/gen_golden.py


def load(name):
  return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def loc_from_lin(ylin, xlin):
  """[H,W,2] float32: (linspace01 - 0.5) on a 'ij' meshgrid (common.py:313-316)."""
  y = (ylin.astype(np.float32) - np.float32(0.5))
  x = (xlin.astype(np.float32) - np.float32(0.5))
  loc = np.empty((y.size, x.size, 2), np.float32)
  loc[..., 0] = y[:, None]
  loc[..., 1] = x[None, :]
  return loc


def f4_inputs(g):
  """(embeddings NCHW, labels or None, grid, ignore or None, iters, loc)."""
  shape = tuple(int(v) for v in g['shape'])
  x = synth.embeddings_nchw(int(g['seed']), shape, str(g['flavour']))
  ign = int(g['ignore'])
  ign = None if ign < 0 else ign
  lab = None
  if bool(g['has_labels']):
    B, _, H, W = shape
    lab = synth.overseg_labels(int(g['label_seed']), B, H, W, regions=48,
                               ignore_rows=4 if ign is not None else 0,
                               ignore_index=255)
    for b in g['fully_ignored']:
      lab[int(b)] = 255
  return x, lab, tuple(int(v) for v in g['grid']), ign, int(g['iters']), \
      loc_from_lin(g['ylin'], g['xlin'])


F4_CASES = ['cfg1_nolabel', 'cfg1_overseg', 'k1_it1', 'mix_overseg', 'c256k64',
            'ragged', 'noignore_labels']
F3_CASES = ['cfg1', 'c256k64', 'c384k128', 'mix']
F19_CASES = ['%s_%s' % (c, f) for c in ('cfg2', 'cfg3', 'cfg4', 'cfg5') for f in ('iid', 'mixture')]
TIE_MARGIN = 1e-6      # float64 top-2 margin below which a label difference against the reference is a near-tie


def f19_case(name):
  """tools/gen_golden.py f19: one whole image of a BASELINE shape.  Returns (g, x NCHW, grid, loc, ref) with
  ref[t] = the REFERENCE's labels after iteration t (t = 0: the grid seeds -- filled in by the caller --, 1 .. 10), and
  forced(t) = what one iteration of the canonical arithmetic gives when started from ref[t - 1]: the reference's
  labels except on the recorded near-tie pixels."""
  g = load('f19_full_' + name)
  shape = tuple(int(v) for v in g['shape'])
  grid = tuple(int(v) for v in g['grid'])
  x = synth.embeddings_nchw(int(g['seed']), shape, str(g['flavour']))
  loc = loc_from_lin(g['ylin'], g['xlin'])
  ref = {1: g['lab1'].astype(np.int64), 2: g['lab2'].astype(np.int64), 9: g['lab9'].astype(np.int64)}
  ref[10] = ref[9].copy()
  ref[10][g['lab10_idx']] = g['lab10_val']
  # iterations 3 .. 8 of the same reference run: deltas in f20 (tools/ref_vs_ref.py, its `base` setting)
  f20 = load('f20_ref_vs_ref')
  for t in range(3, 9):
    ref[t] = ref[t - 1].copy()
    ref[t][f20['%s_lab%d_idx' % (name, t)]] = f20['%s_lab%d_val' % (name, t)]

  def forced(t):
    out = ref[t].copy()
    out[g['tf%d_pixels' % t]] = g['tf%d_oracle' % t]
    return out
  return g, x, grid, loc, ref, forced



def f21_case():
  """tools/gen_golden.py f21: images 0 and 1 of the cfg2 batch with an over-segmentation label map and an ignore
  band, teacher-forced through the REFERENCE's own operator (one iteration from its labels after nine).  Returns
  (g, x, labels, grid, loc, start, want): `start` = the `cluster_indices` argument, `want` = the reference's final
  ids except where a recorded near-tie (float64 margin in g['tie_margin64']) moved a pixel."""
  g = load('f21_full_labelled_cfg2')
  shape = tuple(int(v) for v in g['shape'])
  B, C, H, W = shape
  x = synth.embeddings_nchw(int(g['seed']), shape, 'iid')
  lab = synth.overseg_labels(int(g['label_seed']), B, H, W, regions=48, ignore_rows=4, ignore_index=255)
  want = g['cluster'].astype(np.int64)
  want[g['oracle_cluster_idx']] = g['oracle_cluster_val']
  return (g, x, lab, tuple(int(v) for v in g['grid']), loc_from_lin(g['ylin'], g['xlin']),
          g['start'].astype(np.int64), want)


def check_f21(g, want, emb, emb_loc, labels, cluster, batch):
  assert np.array_equal(labels, g['labels'].astype(np.int64))         # which pixels are kept, in which order
  assert np.array_equal(batch, g['batch'].astype(np.int64))
  assert np.array_equal(cluster, want), '%d ids differ' % int((cluster != want).sum())
  assert g['tie_pixels'].size <= 16 and (g['tie_pixels'].size == 0 or g['tie_margin64'].max() < TIE_MARGIN)
  st = int(g['row_stride'])
  assert np.abs(emb[::st] - g['emb_rows']).max() <= 2e-6
  assert np.abs(emb_loc[::st] - g['emb_loc_rows']).max() <= 2e-6
  n = emb.shape[0]
  assert np.abs(emb.astype(np.float64).sum(0) - g['emb_colsum']).max() <= 2e-7 * n
  assert np.abs(emb_loc.astype(np.float64).sum(0) - g['emb_loc_colsum']).max() <= 2e-7 * n


def exchange_grad_weights(rank, P, C, D):
  """Weights of the per-'GPU' scalar  (protos * w1).sum() + (protos_loc * w2).sum()  whose gradients with respect to
  every GPU's rows the f8 fixture holds (tools/gen_golden.py f8: the reference's own autograd)."""
  w1 = (np.linspace(-1.0, 1.0, P * C, dtype=np.float64).reshape(P, C) * (1.0 + rank)).astype(np.float32)
  w2 = np.cos(np.arange(P * D, dtype=np.float64) * 0.37 + rank).reshape(P, D).astype(np.float32)
  return w1, w2


def check_exchange_grads(g, rank, protos, protos_loc, emb, emb_loc, tol=1e-5):
  """Backward of this rank's scalar through the exchange; the rows' gradients must be the reference's
  (f8: sum over every GPU's replica, i.e. what the all_reduce in the backward carries), <= tol of their scale."""
  import torch
  w1, w2 = exchange_grad_weights(rank, protos.shape[0], protos.shape[1], protos_loc.shape[1])
  dev = protos.device
  ((protos * torch.from_numpy(w1).to(dev)).sum() + (protos_loc * torch.from_numpy(w2).to(dev)).sum()).backward()
  for name, t in (('gemb', emb), ('gloc', emb_loc)):
    want = g['%s%d' % (name, rank)]
    assert t.grad is not None, name
    got = t.grad.detach().cpu().numpy()
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max())
    assert err <= tol * scale, '%s rank %d: |grad - reference| = %.3g of scale %.3g' % (name, rank, err, scale)


def exchange_inputs(seed, n_gpus=2, imgs_per_gpu=3, C=16, K=6):
  """Per-'GPU' pixel sets of the f8_exchange fixture (tools/gen_golden.py f8)."""
  out = []
  for g in range(n_gpus):
    n = 900 + 137 * g
    e = synth.gaussish(seed + 10 * g, n * C).reshape(n, C).astype(np.float64)
    e = (e / np.sqrt((e * e).sum(1, keepdims=True))).astype(np.float32)
    l = synth.gaussish(seed + 10 * g + 1, n * 2).reshape(n, 2) * np.float32(0.3)
    el = np.concatenate([e, l], 1).astype(np.float64)
    el = (el / np.sqrt((el * el).sum(1, keepdims=True))).astype(np.float32)
    img = np.sort((synth.hash_u64(seed + 10 * g + 2, n) % np.uint64(imgs_per_gpu)).astype(np.int64))
    out.append(dict(
        emb=e, emb_loc=el,
        cluster=(synth.hash_u64(seed + 10 * g + 3, n) % np.uint64(K)).astype(np.int64),
        batch=img + imgs_per_gpu * g,
        sem=(synth.hash_u64(seed + 10 * g + 4, n) % np.uint64(4)).astype(np.int64),
        inst=(synth.hash_u64(seed + 10 * g + 5, n) % np.uint64(3)).astype(np.int64),
        image_id=np.array([7, 3, 7, 9, 3, 11][3 * g:3 * g + 3] * 2, np.int64)))
  return out




def set_loss_inputs(seed, n, c, P, nc):
  """Multi-hot labels: every prototype carries 1-3 classes; a pixel carries the classes
  of its own prototype, except every 5th pixel which gets two pseudo-random classes (so
  that some pixels have NO affinity with their own prototype: it then counts as
  'different', loss.py:123-125)."""
  e = synth.gaussish(seed, n * c).reshape(n, c).copy()
  inst = (synth.hash_u64(seed + 1, n) % np.uint64(P)).astype(np.int64)
  h = synth.hash_u64(seed + 2, P * 3).reshape(P, 3)
  psem = np.zeros((P, nc), np.int64)
  for j in range(P):
    for t in range(1 + int(h[j, 0] % np.uint64(3))):
      psem[j, int(h[j, t] % np.uint64(nc))] = 1
  sem = psem[inst].copy()
  hp = synth.hash_u64(seed + 3, n * 2).reshape(n, 2)
  for i in range(0, n, 5):
    sem[i] = 0
    sem[i, int(hp[i, 0] % np.uint64(nc))] = 1
    sem[i, int(hp[i, 1] % np.uint64(nc))] = 1
  return e, inst, sem, psem




def inference_inputs(g):
  """Inputs of tests/golden/f12_inference.npz (tools/gen_golden.py f12)."""
  seed = int(g['seed'])
  n, nk, nc = (int(v) for v in g['maj_shape'])
  sem = (synth.hash_u64(seed, n) % np.uint64(nc)).astype(np.int64)
  clu = (synth.hash_u64(seed + 1, n) % np.uint64(nk)).astype(np.int64)
  clu[clu == 5] = 6
  C, pad_h, pad_w, crop_h, crop_w = (int(v) for v in g['ov_shape'][:5])
  crops, corners, k = [], [], 0
  for ind_h in g['patch_ind_h']:
    for ind_w in g['patch_ind_w']:
      crop = synth.gaussish(seed + 10 + k, C * crop_h * crop_w).reshape(C, crop_h, crop_w).copy()
      if k == 1:
        crop[:, 3, 4] = 0.0
      k += 1
      crops.append(crop)
      corners.append((int(ind_h) - crop_h, int(ind_w) - crop_w))
  protos = synth.gaussish(seed + 3, 11 * 8).reshape(11, 8).copy()
  labs = (synth.hash_u64(seed + 4, 11) % np.uint64(5)).astype(np.int64)
  return sem, clu, crops, corners, (C, pad_h, pad_w), protos, labs


def graph_inputs(seed, B, C, N, K):
  """Inputs of tests/golden/f13_dmon_graph.npz (tools/gen_golden.py f13)."""
  x = synth.gaussish(seed, B * C * N).reshape(B, C, N).copy()
  x /= np.sqrt((x * x).sum(1, keepdims=True))                 # unit columns, like the normalised prototypes
  pad = np.zeros((B, N), bool)
  pad[0, N - 7:] = True
  pad[1, 5] = True
  pad[2, :] = True                                             # an image without valid nodes
  pad[3, 1:] = True                                            # a single valid node: its self loop stays
  seg = (synth.hash_u64(seed + 1, B * N) % np.uint64(2)).astype(np.int64).reshape(B, N)
  seg[1] = 3                                                   # one segment only
  logits = synth.gaussish(seed + 2, B * K * N).reshape(B, K, N).copy()
  return x.astype(np.float32), pad, seg, logits




# ---- one whole training step around a stub backbone ---------------------------
# Restates pyscripts/train/train.py:165-269 for ONE process driving ONE device (lists of
# length 1), with the module set as a parameter: tools/gen_golden.py runs it with the
# reference's modules on CPU, the GPU test with the hsg_amd mirrors.  (train.py itself
# needs cv2 / tensorboardX / the data pipeline and cannot be imported.)
TRAIN_STEP = dict(B=4, C=16, H=20, W=24, grid=(3, 3), iters=5, M=128, KF=6, KC=3, label_divisor=256,
                  ignore=255, kappa=16.0, dmon_knn=3, image_ids=[7, 3, 7, 3])


def train_step_config():
  import types
  c = TRAIN_STEP
  ns = types.SimpleNamespace
  return ns(
      train=ns(img_sim_loss_types='segsort', img_sim_concentration=c['kappa'], img_sim_loss_weight=1.0,
               fine_hrchy_loss_types='segsort', fine_hrchy_concentration=c['kappa'], fine_hrchy_loss_weight=0.5,
               coarse_hrchy_loss_types='segsort', coarse_hrchy_concentration=c['kappa'],
               coarse_hrchy_loss_weight=0.25, dmon_loss_types='dmon', dmon_knn=c['dmon_knn'],
               dmon_loss_weight=0.1, centroid_cont_loss_types='segsort', centroid_cont_concentration=c['kappa'],
               centroid_cont_loss_weight=0.05, fine_hrchy_clusters=c['KF'], coarse_hrchy_clusters=c['KC']),
      dataset=ns(semantic_ignore_index=c['ignore'], num_classes=21),
      network=ns(label_divisor=c['label_divisor']))


def train_step_inputs(seed):
  """numpy inputs of the step: 'backbone output' embeddings [B,C,H,W], position embeddings,
  semantic / instance label maps, image ids (two views per image, not adjacent), and the
  fixed outputs of the two (stubbed) clustering transformers."""
  c = TRAIN_STEP
  B, C, H, W, M, KF, KC = c['B'], c['C'], c['H'], c['W'], c['M'], c['KF'], c['KC']
  x = synth.embeddings_nchw(seed, (B, C, H, W), 'mixture')
  pos = synth.gaussish(seed + 1, B * C * H * W).reshape(B, C, H, W).copy()
  over = synth.overseg_labels(seed + 7, B, H, W, regions=6, ignore_rows=2, ignore_index=c['ignore'])
  sem = np.where(over == c['ignore'], c['ignore'], over % 3).astype(np.int64)
  inst = np.where(over == c['ignore'], 0, over // 3).astype(np.int64)
  Bp = len(set(c['image_ids']))
  g = lambda k, *shape: synth.gaussish(seed + k, int(np.prod(shape))).reshape(shape).copy()
  return dict(x=x, pos=pos, sem=sem, inst=inst, image_id=np.array(c['image_ids'], np.int64),
              fine_logits=g(2, Bp, KF, M) * 2, coarse_logits=g(3, Bp, KC, KF) * 2,
              cent_f=g(4, Bp, C, KF), cent_c=g(5, Bp, C, KC))


def device_inputs(inp, device):
  """The numpy inputs of `run_train_step` as tensors resident on `device`."""
  import torch
  return {k: torch.from_numpy(v).to(device) for k, v in inp.items()}


def run_train_step(mods, inp, device):
  """mods: dict(embedding_cls, prediction_cls, model_utils, loc_fn); returns a dict of tensors
  (losses, accuracy, gradient w.r.t. the embeddings, the integer bookkeeping of the step)."""
  import types
  import torch
  c = TRAIN_STEP
  # (a caller that times the step uploads the batch once -- `device_inputs` -- as a training loop's loader does)
  T = lambda k: inp[k].detach() if torch.is_tensor(inp[k]) else torch.from_numpy(inp[k]).to(device)
  mu = mods['model_utils']
  cfg = train_step_config()
  cent_f, cent_c = T('cent_f').requires_grad_(True), T('cent_c').requires_grad_(True)
  fine_logits, coarse_logits = T('fine_logits').requires_grad_(True), T('coarse_logits').requires_grad_(True)
  emb_cls = mods['embedding_cls']
  stub = types.SimpleNamespace(
      label_divisor=c['label_divisor'], max_num_clusters=c['M'], fine_hrchy_clusters=c['KF'],
      coarse_hrchy_clusters=c['KC'], semantic_ignore_index=c['ignore'],
      kmeans_num_clusters=list(c['grid']), kmeans_iterations=c['iters'],
      fine_query_embed=lambda: None, coarse_query_embed=lambda: None,
      fine_hrchy_transformer=lambda **kw: (cent_f, cent_f * 0.5 + 1.0, fine_logits, kw['src']),
      coarse_hrchy_transformer=lambda **kw: (cent_c, cent_c, coarse_logits, kw['src']))
  for name in ('_calculate_kmeans_prototypes', '_hierarchical_grouping', '_collect_nd_coarser_prototype',
               '_collect_pixel_hierarchical_clustering_indices'):
    setattr(stub, name, types.MethodType(getattr(emb_cls, name), stub))
  x = T('x').requires_grad_(True)
  loc = mods['loc_fn']((c['H'], c['W']), device).unsqueeze(0).expand(c['B'], c['H'], c['W'], 2)

  # train.py:170-174
  image_indices = mu.gather_and_reorder_image_indices([T('image_id')], device)
  label = {'image_index': image_indices[0]}
  # train.py:177 (MultiviewResnetFcn.forward:993-1000 after the backbone)
  emb = emb_cls.generate_clusters(stub, x, T('sem'), T('inst'), label['image_index'], loc, T('pos'))
  # train.py:180-202
  (prototypes, prototypes_with_loc, psem, pinst, pbatch, cluster_indices) = (
      mu.gather_clustering_and_update_prototypes(
          [emb['cluster_embedding']], [emb['cluster_embedding_with_loc']], [emb['cluster_index']],
          [emb['cluster_batch_index']], [emb['cluster_semantic_label']],
          [emb['cluster_instance_label']], device))
  label.update({'prototype': prototypes[0], 'prototype_with_loc': prototypes_with_loc[0],
                'prototype_semantic_label': psem[0], 'prototype_instance_label': pinst[0],
                'prototype_batch_index': pbatch[0]})
  emb['cluster_index'] = cluster_indices[0]
  # train.py:204-228
  for name in ['finehrchy', 'coarsehrchy']:
    inds = torch.gather(label['image_index'], 0, emb['cluster_batch_index'])
    zeros = torch.zeros_like(emb[name + '_cluster_index'])
    protos, protos_loc, _, _, _, c_inds = mu.gather_clustering_and_update_prototypes(
        [emb['cluster_embedding']], [emb['cluster_embedding_with_loc']], [emb[name + '_cluster_index']],
        [inds], [zeros], [zeros], device)
    label[name + '_prototype'] = protos[0]
    label[name + '_prototype_with_loc'] = protos_loc[0]
    emb[name + '_cluster_index'] = c_inds[0]
  # train.py:231-239
  for name in ['finehrchy_', 'coarsehrchy_']:
    label[name + 'mapping_index'] = mu.gather_and_update_cluster_mappings(
        [emb['cluster_index']], [emb[name + 'cluster_index']], device)[0]
  # train.py:244-251
  for key in ['finehrchy_nd_prototype_grouping_centroid', 'coarsehrchy_nd_prototype_grouping_centroid']:
    label[key] = mu.gather_and_update_datas([emb[key].clone()])[0]
  # train.py:260-269
  pred = mods['prediction_cls'](cfg)
  out = pred(emb, label)
  total = out['img_sim_loss'] + out['hrchy_group_loss'] + out['clustering_loss']
  total.backward()
  return dict(img_sim_loss=out['img_sim_loss'].detach(), hrchy_group_loss=out['hrchy_group_loss'].detach(),
              clustering_loss=out['clustering_loss'].detach(), accuracy=out['accuracy'].detach(),
              grad=x.grad, g_fine_logits=fine_logits.grad, g_coarse_logits=coarse_logits.grad,
              g_cent_f=cent_f.grad, image_index=label['image_index'], cluster_index=emb['cluster_index'],
              finehrchy_cluster_index=emb['finehrchy_cluster_index'],
              coarsehrchy_cluster_index=emb['coarsehrchy_cluster_index'],
              finehrchy_mapping_index=label['finehrchy_mapping_index'],
              coarsehrchy_mapping_index=label['coarsehrchy_mapping_index'],
              n_prototypes=torch.tensor(label['prototype'].shape[0]))


# ---- Segsort predictions / losses fixture (tools/gen_golden.py f15) -------------------
def segsort_config():
  import types
  ns = types.SimpleNamespace
  return ns(train=ns(sem_ann_loss_types='segsort', sem_ann_concentration=10.0, sem_ann_loss_weight=1.0,
                     sem_occ_loss_types='segsort', sem_occ_concentration=10.0, sem_occ_loss_weight=0.5,
                     img_sim_loss_types='segsort', img_sim_concentration=10.0, img_sim_loss_weight=0.25,
                     feat_aff_loss_types='none', feat_aff_concentration=10.0, feat_aff_loss_weight=0.0),
            dataset=ns(semantic_ignore_index=255, num_classes=7), network=ns(label_divisor=256))


def segsort_inputs(seed, n=1500, C=24, B=3, K=9, M=60):
  """Pixels of B images in K k-means segments each (ids unique over the batch), 6 semantic classes
  plus an ignored one (255), a memory bank of M labelled prototypes, image-level tags."""
  nc = 7
  g = lambda k, *shape: synth.gaussish(seed + k, int(np.prod(shape))).reshape(shape).copy()
  nrm = lambda a: (a / np.sqrt((a.astype(np.float64) ** 2).sum(1, keepdims=True))).astype(np.float32)
  bidx = np.sort((synth.hash_u64(seed + 1, n) % np.uint64(B)).astype(np.int64))
  cidx = bidx * K + (synth.hash_u64(seed + 2, n) % np.uint64(K)).astype(np.int64)
  seg_sem = (synth.hash_u64(seed + 3, B * K) % np.uint64(nc)).astype(np.int64)
  seg_sem[seg_sem == 6] = 255                                       # an ignored class
  sem = seg_sem[cidx]
  inst = (synth.hash_u64(seed + 4, n) % np.uint64(3)).astype(np.int64)
  cen = nrm(g(5, B * K, C))
  emb = nrm(cen[cidx] + np.float32(0.35) * g(6, n, C))
  emb_loc = nrm(np.concatenate([emb, np.float32(0.2) * g(7, n, 2)], 1))
  protos = nrm(np.stack([emb[cidx == s].sum(0) if (cidx == s).any() else np.ones(C, np.float32)
                         for s in range(B * K)]))
  pbatch = np.repeat(np.arange(B, dtype=np.int64), K)
  tags = np.zeros((B, nc), np.int64)
  for b in range(B):
    for s in seg_sem[b * K:(b + 1) * K]:
      if s < nc:
        tags[b, s] = 1
  mem = nrm(g(8, M, C))
  mem[:B * K] = nrm(protos + np.float32(0.1) * g(9, B * K, C))        # near copies: meaningful retrievals
  mem_lab = (synth.hash_u64(seed + 10, M) % np.uint64(nc - 1)).astype(np.int64)
  return dict(emb=emb, emb_loc=emb_loc, cidx=cidx, sem=sem, inst=inst, bidx=bidx, protos=protos,
              psem=seg_sem, pbatch=pbatch, tags=tags, ptags=tags[pbatch], mem=mem, mem_lab=mem_lab,
              num_classes=np.int64(nc))


def explicit_seed_maps(seed, B, H, W):
  """Per-image initial label maps for `cluster_indices=`: diagonal stripes with a different phase and
  arbitrary (non-dense, partly negative) label values per image, six distinct labels each."""
  yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
  maps = []
  for b in range(B):
    stripe = ((yy + 2 * xx + 5 * b) // 7) % 6
    values = np.array([3, 40, -7, 12, 100 + b, 8], np.int64)           # image-specific, unsorted values
    maps.append(values[stripe])
  return np.stack(maps).astype(np.int64)
