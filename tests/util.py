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


