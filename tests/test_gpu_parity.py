"""GPU parity: libhsgk (through the hsg_amd drop-in surface -> C ABI) against
(1) the golden vectors captured from the reference and (2) the CPU oracle on
the same seeded inputs.  Integer outputs bit-exact; float outputs bit-exact
against the oracle (shared canonical arithmetic) and <= 2e-6 against the
reference's own floats."""
import numpy as np
import pytest

from tests import util
from hsg_amd.utils import synth

pytestmark = pytest.mark.gpu
FTOL = 2e-6


@pytest.fixture(scope='module')
def dev():
  import torch
  assert torch.cuda.is_available(), 'needs a ROCm GPU'
  return torch.device('cuda:0')


def _run_segkm(dev, x, lab, grid, ign, iters, loc=None):
  import torch
  from hsg_amd.utils.segsort import common as sc
  lf = None
  if loc is not None:
    B = x.shape[0]
    lf = torch.from_numpy(loc).to(dev).unsqueeze(0).expand(B, -1, -1, -1)
  out = sc.segment_by_kmeans(
      torch.from_numpy(x).to(dev), None if lab is None else torch.from_numpy(lab).to(dev),
      list(grid), local_features=lf, ignore_index=ign, iterations=iters)
  return [t.cpu().numpy() for t in out]


@pytest.mark.parametrize('case', util.F4_CASES)
def test_segment_by_kmeans_vs_golden_and_oracle(dev, oracle, case):
  g = util.load('f4_segkm_' + case)
  x, lab, grid, ign, iters, loc = util.f4_inputs(g)
  emb, emb_loc, labels, cluster, batch = _run_segkm(dev, x, lab, grid, ign, iters)
  # (1) reference golden vectors
  assert np.array_equal(labels, g['labels'].astype(np.int64))
  assert np.array_equal(batch, g['batch'].astype(np.int64))
  assert np.array_equal(cluster, g['cluster'].astype(np.int64))
  assert np.abs(emb[::util.ROW_STRIDE] - g['emb_rows']).max() <= FTOL
  assert np.abs(emb_loc[::util.ROW_STRIDE] - g['emb_loc_rows']).max() <= FTOL
  # (2) oracle, bit-exact including floats
  ref = oracle.segment_by_kmeans(x, lab, grid, loc, ign, iters)
  for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'),
                        (emb, emb_loc, labels, cluster, batch), ref):
    assert a.shape == b.shape, name
    assert np.array_equal(a, b), '%s: %d mismatching elements' % (name, int((a != b).sum()))


def test_segment_by_kmeans_with_explicit_cluster_indices(dev, oracle):
  """`cluster_indices=` (reference common.py:320-323): the caller's per-image initial labels (arbitrary
  values, made dense per image) instead of the grid seeds -- golden from the reference, bit-exact vs the
  oracle; maps with different label counts per image run group by group."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  g = util.load('f16_segkm_cluster_indices')
  shape = tuple(int(v) for v in g['shape'])
  seed = int(g['seed'])
  x = synth.embeddings_nchw(seed, shape, 'mixture')
  lab = synth.overseg_labels(int(g['label_seed']), shape[0], shape[2], shape[3], regions=5, ignore_rows=2,
                             ignore_index=255)
  ci = util.explicit_seed_maps(seed, shape[0], shape[2], shape[3])
  out = sc.segment_by_kmeans(torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), [9, 9],
                             cluster_indices=torch.from_numpy(ci).to(dev), ignore_index=255,
                             iterations=int(g['iters']))
  emb, emb_loc, labels, cluster, batch = [t.cpu().numpy() for t in out]
  assert np.array_equal(labels, g['labels']) and np.array_equal(batch, g['batch'])
  assert np.array_equal(cluster, g['cluster'])
  assert np.abs(emb[::util.ROW_STRIDE] - g['emb_rows']).max() <= FTOL
  loc = util.loc_from_lin(g['ylin'], g['xlin'])
  ref = oracle.segment_by_kmeans(x, lab, (9, 9), loc, 255, int(g['iters']), cluster_indices=ci)
  for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), (emb, emb_loc, labels, cluster, batch), ref):
    assert np.array_equal(a, b), name
  # maps whose images carry DIFFERENT numbers of distinct labels: one library call per group of images
  # with the same count, batch-wide ids on top -- bit-exact vs the oracle's per-image cluster counts
  uneven = ci.copy()
  uneven[1][uneven[1] == 40] = 3                    # image 1 now has five distinct labels, the others six
  for l, ign in ((lab, 255), (None, None)):
    out = sc.segment_by_kmeans(torch.from_numpy(x).to(dev), None if l is None else torch.from_numpy(l).to(dev), [9, 9],
                               cluster_indices=torch.from_numpy(uneven).to(dev), ignore_index=ign, iterations=3)
    ref = oracle.segment_by_kmeans(x, l, (9, 9), loc, ign, 3, cluster_indices=uneven)
    for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), [t.cpu().numpy() for t in out], ref):
      assert np.array_equal(a, b), name


def test_segment_by_kmeans_huge_label_values_and_many_distinct_labels(dev, oracle):
  """Label values past the library's presence table (>= 2^24, e.g. RGB-packed panoptic ids) and more
  distinct values than it holds: the mirror repeats the call on the ranks of the distinct values (a
  monotone map, so the reference's sorted `unique`s give the same ids) and maps the labels back."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  shape = (2, 128, 24, 40)
  x = synth.embeddings_nchw(77, shape, 'mixture')
  small = synth.overseg_labels(78, shape[0], shape[2], shape[3], regions=6, ignore_rows=2, ignore_index=255)
  big = np.where(small == 255, 255, small.astype(np.int64) * 1000003 + (1 << 30))     # ignore value stays 255
  a = _run_segkm(dev, x, small, (3, 4), 255, 4)
  b = _run_segkm(dev, x, big, (3, 4), 255, 4)
  assert np.array_equal(b[2], a[2].astype(np.int64) * 1000003 + (1 << 30))
  for i in (0, 1, 3, 4):
    assert np.array_equal(a[i], b[i])
  loc = sc._default_loc(shape[2], shape[3], dev).cpu().numpy()     # the operator's own location features
  ref = oracle.segment_by_kmeans(x, big, (3, 4), loc, 255, 4)
  for name, u, v in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), b, ref):
    assert np.array_equal(u, v), name
  # every pixel its own label: far more distinct values than the table's 4096 per (image, cluster)
  many = (np.arange(shape[0] * shape[2] * shape[3], dtype=np.int64).reshape(shape[0], shape[2], shape[3]) * 7919) + (1 << 26)
  c = _run_segkm(dev, x, many, (3, 4), None, 2)
  ref = oracle.segment_by_kmeans(x, many, (3, 4), loc, None, 2)
  for name, u, v in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), c, ref):
    assert np.array_equal(u, v), name


def test_kmeans_grid_seeded_single_map(dev, oracle):
  """`kmeans` (reference common.py:100-126): one [1,H,W,C] map, grid seeds, Lloyd iterations -- equal to the
  oracle's kmeans_with_initial_labels from the same seeds; other batch sizes are rejected (the reference
  pairs H*W seed labels with B*H*W rows)."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  H, W, C = 24, 36, 130
  x = synth.gaussish(4711, H * W * C).reshape(1, H, W, C)
  x = (x / np.sqrt((x.astype(np.float64) ** 2).sum(-1, keepdims=True))).astype(np.float32)
  got = sc.kmeans(torch.from_numpy(x).to(dev), [3, 4], iterations=6).cpu().numpy()
  seeds = sc.initialize_cluster_labels([3, 4], [H, W], 'cpu').numpy().reshape(-1)
  ref = oracle.kmeans_with_initial_labels(x.reshape(-1, C), seeds, int(seeds.max()) + 1, 6)
  assert got.shape == (1, H, W) and np.array_equal(got.reshape(-1), ref)
  with pytest.raises(ValueError):
    sc.kmeans(torch.from_numpy(np.concatenate([x, x])).to(dev), [3, 4])


def test_explicit_local_features_match_default(dev):
  g = util.load('f4_segkm_ragged')
  x, lab, grid, ign, iters, loc = util.f4_inputs(g)
  a = _run_segkm(dev, x, lab, grid, ign, iters)
  b = _run_segkm(dev, x, lab, grid, ign, iters, loc=loc)
  for u, v in zip(a, b):
    assert np.array_equal(u, v)


@pytest.mark.parametrize('case', util.F3_CASES)
def test_kmeans_with_initial_labels(dev, oracle, case):
  import torch
  from hsg_amd.utils.segsort import common as sc
  g = util.load('f3_kmeans_' + case)
  shape = tuple(int(v) for v in g['shape'])
  grid = tuple(int(v) for v in g['grid'])
  x = synth.embeddings_nchw(int(g['seed']), shape, str(g['flavour']))
  B, C, H, W = shape
  loc = util.loc_from_lin(g['ylin'], g['xlin'])
  _, emb_loc, _, _, _ = oracle.segment_by_kmeans(x, None, grid, loc, None, 0)
  seeds = oracle.dense_relabel(oracle.initialize_cluster_labels(grid, (H, W)).reshape(-1))
  K = int(g['K'])
  for b in range(B):
    rows = np.ascontiguousarray(emb_loc[b * H * W:(b + 1) * H * W])
    for it in (1, 2, 10, 15):
      lab = sc.kmeans_with_initial_labels(
          torch.from_numpy(rows).to(dev), torch.from_numpy(seeds).to(dev), K, it)
      lab = lab.cpu().numpy()
      assert np.array_equal(lab, g['b%d_it%d' % (b, it)].astype(np.int64)), (case, b, it)


@pytest.mark.parametrize('n,d,K', [(1, 4, 1), (63, 7, 3), (2049, 34, 8), (5000, 258, 64),
                                   (3000, 130, 100), (777, 66, 256)])
def test_find_nearest_prototypes_vs_oracle(dev, oracle, n, d, K):
  import torch
  from hsg_amd.utils.segsort import common as sc
  x = oracle.normalize_embedding(synth.gaussish(11 + n, n * d).reshape(n, d))
  p = oracle.normalize_embedding(synth.gaussish(13 + K, K * d).reshape(K, d))
  if K > 2:
    p[1] = p[0]                      # exact tie -> first index must win
    p[K - 1] = 0.0                   # empty-cluster style zero prototype
  got = sc.find_nearest_prototypes(torch.from_numpy(x).to(dev), torch.from_numpy(p).to(dev))
  ref = oracle.find_nearest_prototypes(x, p)
  assert np.array_equal(got.cpu().numpy(), ref)


def test_normalize_embedding_bit_exact(dev, oracle):
  import torch
  from hsg_amd.utils.general import common as gc
  x = synth.gaussish(5, 300 * 70).reshape(300, 70).copy()
  x[3] = 0.0
  x[7] *= np.float32(1e-20)
  y = gc.normalize_embedding(torch.from_numpy(x).to(dev)).cpu().numpy()
  assert np.array_equal(y, oracle.normalize_embedding(x))
  g = util.load('f1_normalize')
  n, d = (int(v) for v in g['shape'])
  x = synth.gaussish(int(g['seed']), n * d).reshape(n, d).copy()
  x[3] = 0.0
  x[7] *= np.float32(1e-20)
  x[11] *= np.float32(1e-9)
  y = gc.normalize_embedding(torch.from_numpy(x).to(dev)).cpu().numpy()
  assert np.abs(y - g['y']).max() <= FTOL


def test_prototypes_and_segment_mean_vs_golden_and_oracle(dev, oracle):
  import torch
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.general import common as gc
  g = util.load('f5_prototypes')
  n, d = int(g['n']), int(g['d'])
  x = oracle.normalize_embedding(synth.gaussish(int(g['seed']), n * d).reshape(n, d))
  lab = g['labels'].astype(np.int64)
  xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
  p_auto = sc.calculate_prototypes_from_labels(xt, lt).cpu().numpy()
  p_pad = sc.calculate_prototypes_from_labels(xt, lt, 64).cpu().numpy()
  sm = gc.segment_mean(xt, lt).cpu().numpy()
  assert np.abs(p_auto - g['p_auto']).max() <= FTOL
  assert np.abs(p_pad - g['p_pad']).max() <= FTOL
  assert np.abs(sm - g['seg_mean']).max() <= FTOL
  assert np.array_equal(p_auto, oracle.calculate_prototypes_from_labels(x, lab))
  assert np.array_equal(p_pad, oracle.calculate_prototypes_from_labels(x, lab, 64))
  assert np.array_equal(sm, oracle.segment_mean(x, lab))
  assert np.all(p_pad[5] == 0) and np.all(p_pad[37:] == 0)


@pytest.mark.parametrize('n,d,P', [(1, 3, 1), (2047, 34, 7), (2049, 258, 64), (30000, 130, 700),
                                   (9000, 258, 300), (5000, 66, 1000), (6000, 386, 128), (4100, 1000, 50),
                                   (3000, 1100, 20)])          # (d > 1024: the LDS-table kernels)
def test_segment_reduce_sorted_ids_vs_oracle(dev, oracle, n, d, P):
  """Large P with image-major style (monotone, locally clustered) ids."""
  import torch
  from hsg_amd import ops
  x = oracle.normalize_embedding(synth.gaussish(7 + n, n * d).reshape(n, d))
  base = (np.arange(n, dtype=np.int64) * P) // n               # monotone 0..P-1
  jitter = (synth.hash_u64(3 + n, n) % np.uint64(5)).astype(np.int64)
  lab = np.clip(base + jitter - 2, 0, P - 1)
  lab[::97] = -1                                               # skipped rows
  xt, lt = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
  for mode in (0, 1, 2):
    got = ops.segment_reduce(xt, lt, P, mode).cpu().numpy()
    if mode == 0:
      ref = oracle.calculate_prototypes_from_labels(x, lab, P)
    else:
      import ctypes
      ref = np.empty((P, d), np.float32)
      oracle.lib().orc_segment_sums(
          x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_int64(n), d,
          lab.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int64(P), oracle.CHUNK,
          ref.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
      if mode == 1:
        cnt = np.bincount(lab[lab >= 0], minlength=P).astype(np.float32)
        cnt[cnt == 0] = 1
        ref = ref / cnt[:, None]
    assert np.array_equal(got, ref), (mode, int((got != ref).sum()))


@pytest.mark.parametrize('n,d,P,pattern', [
    (30000, 34, 3000, 'random'),        # ~1500 distinct ids per 2048-row chunk: per-segment scan path
    (9000, 258, 5000, 'random'),
    (20000, 66, 40000, 'stride'),       # few distinct ids per chunk, id range 30000+: rank path
    (20000, 130, 200000, 'huge'),       # id range beyond the bitmap: scan path
    (6000, 20, 4000, 'small_images'),   # several small images per chunk, label-split clusters
])
def test_segment_reduce_arbitrary_ids_vs_oracle(dev, oracle, n, d, P, pattern):
  """Scattered segment ids (the reference's scatter_add_ accepts any labels): more than 512
  distinct ids per chunk, wide id ranges with few distinct ids, ranges beyond the bitmap --
  all bit-exact vs the oracle's order C2, forward and backward."""
  import ctypes
  import torch
  from hsg_amd import ops
  x = synth.gaussish(11 + n, n * d).reshape(n, d).copy()
  h = synth.hash_u64(5 + n, n)
  if pattern == 'random':
    lab = (h % np.uint64(P)).astype(np.int64)
  elif pattern == 'stride':
    lab = ((np.arange(n) // 500) * 997 + (h % np.uint64(3)).astype(np.int64) * 9973) % P
  elif pattern == 'huge':
    lab = ((h % np.uint64(40)).astype(np.int64) * 49999 + 7) % P
  else:
    lab = (np.arange(n) // 300) * 200 + (h % np.uint64(190)).astype(np.int64)
    lab = np.minimum(lab, P - 1)
  lab = lab.astype(np.int64)
  xt = torch.from_numpy(x).to(dev).requires_grad_(True)
  lt = torch.from_numpy(lab).to(dev)
  ref = np.empty((P, d), np.float32)
  oracle.lib().orc_segment_sums(
      x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_int64(n), d,
      lab.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int64(P), oracle.CHUNK,
      ref.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
  got = ops.segment_reduce(xt, lt, P, 2)
  assert np.array_equal(got.detach().cpu().numpy(), ref), int((got.detach().cpu().numpy() != ref).sum())
  w = torch.from_numpy(synth.gaussish(13 + n, P * d).reshape(P, d).copy()).to(dev)
  (got * w).sum().backward()
  assert torch.equal(xt.grad, w[lt])
  protos = ops.segment_reduce(xt.detach(), lt, P, 0).cpu().numpy()
  assert np.array_equal(protos, oracle.calculate_prototypes_from_labels(x, lab, P))


def test_segment_reduce_out_of_range_label_is_reported(dev, monkeypatch):
  """calculate_prototypes_from_labels mirrors the reference's scatter_add_: a label outside [0, max_label) is an
  error raised BY THE OFFENDING CALL with no environment switch set (Python exceptions at the call);
  HSGK_SYNC_ERRORS=0 keeps the host out of it and a later call reports."""
  import torch
  from hsg_amd import _lib, ops
  from hsg_amd.utils.segsort import common as sc
  x = torch.randn((100, 8), device=dev)
  lab = torch.arange(100, device=dev) % 7
  lab[13] = 9
  monkeypatch.delenv('HSGK_SYNC_ERRORS', raising=False)
  with pytest.raises(_lib.HsgkError):
    sc.calculate_prototypes_from_labels(x, lab, max_label=7)        # at the call
  _lib.poll_deferred(wait=True)                                     # nothing left to report
  neg = lab.clone()
  neg[13] = -1
  with pytest.raises(_lib.HsgkError):
    sc.calculate_prototypes_from_labels(x, neg)                     # default max_label: the read it does anyway
  assert sc.calculate_prototypes_from_labels(x, lab).shape == (10, 8)
  monkeypatch.setenv('HSGK_SYNC_ERRORS', '0')
  sc.calculate_prototypes_from_labels(x, lab, max_label=7)          # flag travels behind the kernels
  with pytest.raises(_lib.HsgkError):
    _lib.poll_deferred(wait=True)
  _lib.poll_deferred(wait=True)                                     # reported once
  monkeypatch.delenv('HSGK_SYNC_ERRORS')
  ops.segment_reduce(x, lab, 7, 0, strict=True)                     # the library-level op: deferred unless asked
  with pytest.raises(_lib.HsgkError):
    _lib.poll_deferred(wait=True)
  monkeypatch.setenv('HSGK_SYNC_ERRORS', '1')
  with pytest.raises(_lib.HsgkError):
    ops.segment_reduce(x, lab, 7, 0, strict=True)


def test_prototype_gradients_match_torch_autograd(dev):
  """Backward of calculate_prototypes_from_labels / segment_mean against a
  plain torch (ATen, fp32) restatement of the same op on the GPU."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.general import common as gc
  n, d, P = 4000, 66, 37
  x = torch.from_numpy(synth.gaussish(21, n * d).reshape(n, d)).to(dev)
  lab = torch.from_numpy((synth.hash_u64(22, n) % np.uint64(P)).astype(np.int64)).to(dev)
  w = torch.from_numpy(synth.gaussish(23, P * d).reshape(P, d)).to(dev)

  def ref_proto(xx):
    acc = torch.zeros((P, d), device=dev).index_add_(0, lab, xx)
    nrm = acc.norm(dim=1, keepdim=True)
    return acc / torch.where(nrm >= 1e-12, nrm, torch.full_like(nrm, 1e-12))

  def ref_mean(xx):
    acc = torch.zeros((P, d), device=dev).index_add_(0, lab, xx)
    cnt = torch.bincount(lab, minlength=P).clamp(min=1).float()
    return acc / cnt[:, None]

  for ours, ref in ((lambda t: sc.calculate_prototypes_from_labels(t, lab, P), ref_proto),
                    (lambda t: gc.segment_mean(t, lab), ref_mean)):
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    (ours(a) * w).sum().backward()
    (ref(b) * w).sum().backward()
    scale = b.grad.abs().max().item()
    assert (a.grad - b.grad).abs().max().item() <= 1e-5 * max(scale, 1.0)


def test_segsort_loss_vs_golden_and_oracle(dev, oracle):
  """Loss value within 1e-4 (north_star), per-pixel nll and gradients vs the
  reference's autograd captured in tests/golden/f6_segsort_loss.npz."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort.loss import SegSortLoss
  g = util.load('f6_segsort_loss')
  n, c, P = int(g['n']), int(g['c']), int(g['P'])
  e_np = oracle.normalize_embedding(synth.gaussish(int(g['seed']), n * c).reshape(n, c))
  inst = torch.from_numpy(g['inst'].astype(np.int64)).to(dev)
  psem = torch.from_numpy(g['psem'].astype(np.int64)).to(dev)
  sem = psem[inst]
  for kappa in (10, 16):
    for mode, tag in (('segsort+', 'plus'), ('segsort', 'plain')):
      key = 'k%d_%s' % (kappa, tag)
      e = torch.from_numpy(e_np).to(dev).requires_grad_(True)
      proto = sc.calculate_prototypes_from_labels(e, inst, P)
      assert np.abs(proto.detach().cpu().numpy() - g['proto']).max() <= FTOL
      pp = proto.detach().clone().requires_grad_(True)
      loss = SegSortLoss(kappa, mode)(e, sem, inst, pp, psem)
      loss.backward()
      assert abs(loss.item() - float(g[key + '_loss'])) <= 1e-4
      nll = SegSortLoss(kappa, mode, reduction='none')(e.detach(), sem, inst, proto.detach(), psem)
      assert nll.shape == (n, 1)
      assert np.abs(nll.view(-1).cpu().numpy() - g[key + '_nll']).max() <= 1e-4
      assert np.abs(e.grad.cpu().numpy()[::7] - g[key + '_gemb']).max() <= 2e-6
      assert np.abs(pp.grad.cpu().numpy() - g[key + '_gproto']).max() <= 2e-5
      ref = oracle.segsort_nll(e_np, sem.cpu().numpy(), inst.cpu().numpy(),
                               proto.detach().cpu().numpy(), psem.cpu().numpy(), float(kappa), mode)
      assert abs(loss.item() - ref.mean()) <= 1e-4


def _ref_nll_torch(emb, sem, inst, proto, psem, kappa, plus):
  """loss.py:15-82 restated in float64 ATen (the reference formula, materialised)."""
  import torch
  sim = torch.exp(emb.double() @ proto.double().t() * kappa)
  own = sim.gather(1, inst.view(-1, 1))
  same = (sem.view(-1, 1) == psem.view(1, -1)).double()
  num = own
  if plus:
    sw = (sim * same).sum(1, keepdim=True) - own
    num = torch.where(sw > 0, sw, own)
  den = (sim * (1.0 - same)).sum(1, keepdim=True) + num
  return -(num / den).log().view(-1)


@pytest.mark.parametrize('n,c,P', [(700, 32, 40), (3000, 48, 333), (1500, 130, 97), (4500, 256, 700),
                                   (2100, 384, 260), (129, 20, 3), (40, 256, 1500)])
def test_segsort_losses_three_label_sets_one_pass_fwd_bwd(dev, oracle, n, c, P):
  """`segsort_losses` (one E P^T pass, three label sets with different labels, concentrations
  and group modes -- hsg/models/predictions/hsg.py:78-155) against the reference formula in
  float64: every loss within 1e-4, the gradients of a weighted sum w.r.t. embeddings and
  prototypes (streaming backward: score tiles recomputed and contracted in place, no [N,P]
  storage) within 1e-5 of their scale; and == three single-set SegSortLoss calls."""
  import torch
  from hsg_amd.utils.segsort.loss import SegSortLoss, segsort_losses
  e_np = oracle.normalize_embedding(synth.gaussish(71 + n, n * c).reshape(n, c))
  p_np = oracle.normalize_embedding(synth.gaussish(72 + n, P * c).reshape(P, c))
  inst = torch.from_numpy((synth.hash_u64(73 + n, n) % np.uint64(P)).astype(np.int64)).to(dev)
  sets = []
  # ('segsort+' subtracts the own similarity from the same-label sum in fp32, loss.py:63-66: with few
  #  channels the random cosines spread widely and exp(16 cos) of the own prototype can dwarf the rest
  #  of that sum, whose fp32 rounding the float64 reference below does not share -- smaller
  #  concentrations there keep the comparison about the kernel, not about that cancellation)
  k_hi, k_lo = (16.0, 10.0) if c >= 32 else (4.0, 3.0)
  for i, (classes, kappa, mode) in enumerate(((5, k_hi, 'segsort+'), (max(P // 3, 1), k_lo, 'segsort'),
                                              (2, k_hi, 'segsort+'))):
    psem = torch.from_numpy((synth.hash_u64(80 + i + n, P) % np.uint64(classes)).astype(np.int64)).to(dev)
    sem = psem[inst].clone()
    flip = torch.from_numpy((synth.hash_u64(90 + i + n, n) % np.uint64(7) == 0)).to(dev)
    sem[flip] = (sem[flip] + 1) % classes          # pixels whose label differs from their own prototype's
    sets.append((sem, psem, kappa, mode))
  wts = (1.0, 0.5, 2.0)
  e = torch.from_numpy(e_np).to(dev).requires_grad_(True)
  pr = torch.from_numpy(p_np).to(dev).requires_grad_(True)
  losses = segsort_losses(e, inst, pr, sets)
  sum(w * l for w, l in zip(wts, losses)).backward()
  e2 = torch.from_numpy(e_np).to(dev).requires_grad_(True)
  p2 = torch.from_numpy(p_np).to(dev).requires_grad_(True)
  refs = [_ref_nll_torch(e2, s, inst, p2, ps, k, m == 'segsort+').mean() for s, ps, k, m in sets]
  sum(w * l for w, l in zip(wts, refs)).backward()
  for a, b in zip(losses, refs):
    assert abs(a.item() - b.item()) <= 1e-4 * max(1.0, abs(b.item())), (a.item(), b.item())
  for got, ref in ((e.grad, e2.grad), (pr.grad, p2.grad)):
    scale = max(ref.abs().max().item(), 1e-6)
    assert (got.double() - ref.double()).abs().max().item() <= 1e-5 * scale + 1e-9, \
        ((got.double() - ref.double()).abs().max().item(), scale)
  single = [SegSortLoss(k, m)(e.detach(), s, inst, pr.detach(), ps) for s, ps, k, m in sets]
  for a, b in zip(losses, single):
    assert a.item() == b.item()


def test_segsort_loss_large_shapes_vs_oracle(dev, oracle):
  """C=256 pixels against several prototype blocks incl. a ragged last block
  and a multi-chunk pixel range."""
  import torch
  from hsg_amd.utils.segsort.loss import SegSortLoss
  n, c, P = 5000, 256, 150
  e = oracle.normalize_embedding(synth.gaussish(91, n * c).reshape(n, c))
  p = oracle.normalize_embedding(synth.gaussish(92, P * c).reshape(P, c))
  inst = (synth.hash_u64(93, n) % np.uint64(P)).astype(np.int64)
  psem = (synth.hash_u64(94, P) % np.uint64(11)).astype(np.int64)
  sem = psem[inst]
  t = lambda a: torch.from_numpy(a).to(dev)
  for mode in ('segsort+', 'segsort'):
    nll = SegSortLoss(16, mode, reduction='none')(t(e), t(sem), t(inst), t(p), t(psem))
    ref = oracle.segsort_nll(e, sem, inst, p, psem, 16.0, mode)
    assert np.abs(nll.view(-1).cpu().numpy() - ref).max() <= 1e-4


@pytest.mark.parametrize('case', ['ragged', 'cfg1_overseg', 'c256k64'])
def test_segment_by_kmeans_backward_vs_torch_autograd(dev, case):
  """Gradients of the two float outputs w.r.t. the NCHW input against a plain
  torch fp32 restatement (normalise -> cat -> normalise -> index_select)."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  g = util.load('f4_segkm_' + case)
  x, lab, grid, ign, iters, loc = util.f4_inputs(g)
  B, C, H, W = x.shape
  xt = torch.from_numpy(x).to(dev)
  lt = None if lab is None else torch.from_numpy(lab).to(dev)
  a = xt.clone().requires_grad_(True)
  emb, eloc, labels, cluster, batch = sc.segment_by_kmeans(a, lt, list(grid), ignore_index=ign,
                                                           iterations=2)
  n = emb.shape[0]
  w1 = torch.from_numpy(synth.gaussish(5, n * C).reshape(n, C)).to(dev)
  w2 = torch.from_numpy(synth.gaussish(6, n * (C + 2)).reshape(n, C + 2)).to(dev)
  ((emb * w1).sum() + (eloc * w2).sum()).backward()

  b = xt.clone().requires_grad_(True)
  e = b.permute(0, 2, 3, 1).reshape(-1, C)
  e = e / e.norm(dim=1, keepdim=True).clamp_min(1e-12)
  lc = torch.from_numpy(loc).to(dev).view(1, H * W, 2).expand(B, H * W, 2).reshape(-1, 2)
  el = torch.cat([e, lc], 1)
  el = el / el.norm(dim=1, keepdim=True).clamp_min(1e-12)
  if ign is not None and lt is not None:
    keep = (lt.view(-1) != ign).nonzero().view(-1)
    e, el = e.index_select(0, keep), el.index_select(0, keep)
  assert e.shape[0] == n
  ((e * w1).sum() + (el * w2).sum()).backward()
  scale = b.grad.abs().max().item()
  assert (a.grad - b.grad).abs().max().item() <= 2e-5 * max(scale, 1.0)
  assert not labels.requires_grad and not cluster.requires_grad


@pytest.mark.parametrize('shape,ignore_rows', [((2, 384, 21, 37), 3), ((1, 128, 40, 33), 0), ((2, 512, 9, 70), 2),
                                               ((1, 320, 17, 19), 0)])
def test_prep_backward_wide_rows_vs_torch_autograd(dev, shape, ignore_rows):
  """The 32-pixel backward kernel with one / two quads per lane (C up to 512), ragged half tiles and
  an ignore band, against torch autograd of the plain restatement."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  B, C, H, W = shape
  x = synth.embeddings_nchw(77 + C, shape, 'iid')
  lab = synth.overseg_labels(78, B, H, W, regions=5, ignore_rows=ignore_rows) if ignore_rows else None
  ign = 255 if ignore_rows else None
  xt = torch.from_numpy(x).to(dev)
  lt = None if lab is None else torch.from_numpy(lab).to(dev)
  a = xt.clone().requires_grad_(True)
  emb, eloc, _, _, _ = sc.segment_by_kmeans(a, lt, [3, 2], ignore_index=ign, iterations=1)
  n = emb.shape[0]
  w1 = torch.from_numpy(synth.gaussish(5, n * C).reshape(n, C)).to(dev)
  w2 = torch.from_numpy(synth.gaussish(6, n * (C + 2)).reshape(n, C + 2)).to(dev)
  ((emb * w1).sum() + (eloc * w2).sum()).backward()
  b = xt.clone().requires_grad_(True)
  e = b.permute(0, 2, 3, 1).reshape(-1, C)
  e = e / e.norm(dim=1, keepdim=True).clamp_min(1e-12)
  loc = (sc.generate_location_features((H, W), dev, 'float') - 0.5).view(1, H * W, 2).expand(B, H * W, 2).reshape(-1, 2)
  el = torch.cat([e, loc], 1)
  el = el / el.norm(dim=1, keepdim=True).clamp_min(1e-12)
  if lt is not None:
    keep = (lt.view(-1) != ign).nonzero().view(-1)
    e, el = e.index_select(0, keep), el.index_select(0, keep)
  assert e.shape[0] == n
  ((e * w1).sum() + (el * w2).sum()).backward()
  scale = b.grad.abs().max().item()
  assert (a.grad - b.grad).abs().max().item() <= 2e-5 * max(scale, 1.0)


def test_exchange_list_api_on_gpu_vs_reference(dev):
  """hsg_amd.models.utils with the libhsgk kernels (segment sums, normalise)
  against the reference's outputs for the two-'GPU' fixture."""
  import torch
  from hsg_amd.models import utils as mu
  g = util.load('f8_exchange')
  parts = util.exchange_inputs(int(g['seed']))
  T = lambda k: [torch.from_numpy(p[k]).to(dev) for p in parts]
  embs = [t.requires_grad_(True) for t in T('emb')]
  protos, protos_loc, psem, pinst, pbatch, upd = mu.gather_clustering_and_update_prototypes(
      embs, T('emb_loc'), T('cluster'), T('batch'), T('sem'), T('inst'), dev)
  assert np.array_equal(psem[0].cpu().numpy(), g['psem'])
  assert np.array_equal(pinst[0].cpu().numpy(), g['pinst'])
  assert np.array_equal(pbatch[0].cpu().numpy(), g['pbatch'])
  assert np.array_equal(upd[0].cpu().numpy(), g['upd0'])
  assert np.array_equal(upd[1].cpu().numpy(), g['upd1'])
  assert np.abs(protos[0].detach().cpu().numpy() - g['protos']).max() <= FTOL
  assert np.abs(protos_loc[1].detach().cpu().numpy() - g['protos_loc']).max() <= FTOL
  protos[0].sum().backward()
  assert embs[0].grad is not None and embs[1].grad is not None


def _torch_segsort_nll(e, sem, inst, p, psem, kappa, plus):
  """loss.py:15-82 in plain PyTorch on the device (the [N, P] matrix is materialised), in the dtype of `e`."""
  import torch
  s = torch.exp(torch.mm(e, p.t()) * kappa)
  same = (sem.view(-1, 1) == psem.view(1, -1)).to(e.dtype)
  own = torch.gather(s, 1, inst.view(-1, 1)).view(-1)
  same_sum = (s * same).sum(1)
  diff = (s * (1.0 - same)).sum(1)
  num = own
  if plus:
    wo = same_sum - own
    num = torch.where(wo > 0, wo, own)
  return -torch.log(num / (num + diff))


@pytest.mark.parametrize('n,c,P', [(9408, 128, 1536), (37632, 256, 3072), (200704, 256, 3072)])
@pytest.mark.parametrize('engine', ['split', 'fp32'])
def test_segsort_loss_at_training_and_benchmark_scale(dev, oracle, monkeypatch, n, c, P, engine):
  """SegSortLoss where the reference runs it (predictions/hsg.py:105,130,149: N = 9-40 K pixels per GPU against
  P = 1.5-6 K prototypes) and at one benchmark image (N = 200 704): loss within 1e-4 and both gradients against
  plain PyTorch fp32 on the device, per-pixel values of a pixel sample against the oracle -- for the bf16x3
  forward engine (default) and the fp32 one (HSGK_LOSS=fp32)."""
  import torch
  from hsg_amd.utils.segsort import loss as sl
  monkeypatch.setenv('HSGK_LOSS', engine)
  g = torch.Generator(device=dev).manual_seed(n + P)
  proto = torch.nn.functional.normalize(torch.randn((P, c), device=dev, generator=g), dim=1)
  inst = torch.randint(0, P, (n,), device=dev, generator=g)
  e = torch.nn.functional.normalize(proto[inst] + 0.35 * torch.randn((n, c), device=dev, generator=g), dim=1)
  psem = torch.arange(P, device=dev) % 21
  sem = psem[inst].clone()
  flip = torch.rand((n,), device=dev, generator=g) < 0.1
  sem[flip] = (sem[flip] + 3) % 21
  for kappa, mode in ((16.0, 'segsort+'), (10.0, 'segsort')):
    # float64 reference: in fp32 the 'segsort+' numerator same - own cancels where the own prototype dominates
    # (loss.py:63-66), so an fp32 formulation is its own noise source
    e2, p2 = e.double().requires_grad_(True), proto.double().requires_grad_(True)
    ref_nll = _torch_segsort_nll(e2, sem, inst, p2, psem, kappa, mode == 'segsort+')
    # 'segsort+': numerator = (same-label sum) - own (loss.py:63-66), computed in fp32 like the reference does.
    # Where the own similarity dominates that sum, or -- own prototype with another label -- the two unrelated
    # sums nearly agree, the difference amplifies the rounding of the scores by cond = (same + own) / |same - own|
    # (the reference's own fp32 value included).  Per pixel the tolerance scales with cond; badly conditioned
    # pixels are left out of the gradient comparison.
    with torch.no_grad():
      sd = torch.exp(torch.mm(e.double(), proto.double().t()) * kappa)
      own = torch.gather(sd, 1, inst.view(-1, 1)).view(-1)
      same = (sd * (sem.view(-1, 1) == psem.view(1, -1))).sum(1)
      cond = (same + own) / (same - own).abs() if mode == 'segsort+' else torch.ones_like(own)
      well = cond < 20.0
    assert float(well.float().mean()) > 0.5
    wts = well.double() / well.sum()
    et, pt = e.clone().requires_grad_(True), proto.clone().requires_grad_(True)
    nll = sl.segsort_nll(et, sem, inst, pt, psem, kappa, mode)
    assert abs(nll.mean().item() - ref_nll.mean().item()) <= 1e-4, (nll.mean().item(), ref_nll.mean().item())
    assert ((nll.double() - ref_nll).abs() <= 1e-5 * cond.clamp_min(10.0)).all()
    (nll * wts.float()).sum().backward()
    (ref_nll * wts).sum().backward()
    for a, b in ((et.grad, e2.grad), (pt.grad, p2.grad)):
      assert (a.double() - b).abs().max().item() <= 3e-5 * max(b.abs().max().item(), 1e-9) + 1e-12
    # a sample of pixels against the oracle (float64 arithmetic on the fp32 inputs)
    idx = torch.arange(0, n, max(1, n // 1500), device=dev)
    got = nll.detach()[idx].cpu().numpy()
    want = oracle.segsort_nll(e[idx].cpu().numpy(), sem[idx].cpu().numpy(), inst[idx].cpu().numpy(),
                              proto.cpu().numpy(), psem.cpu().numpy(), kappa, mode).reshape(-1)
    assert (np.abs(got - want) <= 1e-5 * np.maximum(10.0, cond[idx].cpu().numpy())).all()


@pytest.mark.parametrize('c', [64, 128, 256])
@pytest.mark.parametrize('route', ['h16', 'mixed', 'fast32', 'generic'])
def test_loss_backward_routes_vs_float64(dev, oracle, monkeypatch, c, route):
  """The four backward tiles of loss.hip -- both contractions on the fp16 pipe (default), fp16 scores + fp32
  second contraction (HSGK_LOSS_BWD=mixed), all fp32 with wide operand reads (HSGK_LOSS=fp32), the general tile
  (HSGK_LOSS_BWD=generic) -- for three label sets in one pass and for a grouped single set, ragged against the
  16- / 32-row blocks, with per-pixel upstream gradients spanning six decades (the fp16 W operand is scaled per
  owner pixel / by the launch maximum) and some exactly zero: both gradients against float64 autograd of the
  reference formula."""
  import torch
  from hsg_amd.utils.segsort import loss as sl
  if route == 'mixed': monkeypatch.setenv('HSGK_LOSS_BWD', 'mixed')
  if route == 'generic': monkeypatch.setenv('HSGK_LOSS_BWD', 'generic')
  if route == 'fast32': monkeypatch.setenv('HSGK_LOSS', 'fp32')
  n, P = 2531, 333
  e_np = oracle.normalize_embedding(synth.gaussish(171 + c, n * c).reshape(n, c))
  p_np = oracle.normalize_embedding(synth.gaussish(172 + c, P * c).reshape(P, c))
  T = lambda a: torch.from_numpy(a).to(dev)
  inst = T((synth.hash_u64(173 + c, n) % np.uint64(P)).astype(np.int64))
  sets = []
  for i, (classes, kappa, mode) in enumerate(((5, 16.0, 'segsort+'), (P // 3, 10.0, 'segsort'), (2, 16.0, 'segsort+'))):
    psem = T((synth.hash_u64(180 + i + c, P) % np.uint64(classes)).astype(np.int64))
    sem = psem[inst].clone()
    flip = T((synth.hash_u64(190 + i + c, n) % np.uint64(7) == 0))
    sem[flip] = (sem[flip] + 1) % classes
    sets.append((sem, psem, kappa, mode))
  # upstream gradient per pixel: 1e-3 .. 1e3, every 11th pixel exactly zero
  up = T(np.exp(np.log(10.0) * 3.0 * synth.gaussish(199 + c, n).clip(-1, 1)).astype(np.float32))
  up[::11] = 0.0
  # 'segsort+' pixels whose fp32 numerator same - own cancels (loss.py:63-66) are rounding noise in the reference
  # itself: no upstream gradient for them, as in the scale test above
  with torch.no_grad():
    for sem_, psem_, kappa_, mode_ in sets:
      if mode_ != 'segsort+': continue
      for kk in (kappa_, 12.0):
        sd = torch.exp(torch.mm(T(e_np).double(), T(p_np).double().t()) * kk)
        own_ = torch.gather(sd, 1, inst.view(-1, 1)).view(-1)
        same_ = (sd * (sem_.view(-1, 1) == psem_.view(1, -1))).sum(1)
        up[(same_ + own_) / (same_ - own_).abs() >= 20.0] = 0.0
  assert float((up > 0).float().mean()) > 0.4
  for nsets in (1, 3):
    e, pr = T(e_np).requires_grad_(True), T(p_np).requires_grad_(True)
    nll = sl.segsort_losses(e, inst, pr, sets[:nsets], reduction='none')
    sum((l.view(-1) * up).sum() * w for l, w in zip(nll, (1.0, 0.5, 2.0))).backward()
    e2, p2 = T(e_np).double().requires_grad_(True), T(p_np).double().requires_grad_(True)
    refs = [_ref_nll_torch(e2, s_, inst, p2, ps, k, m == 'segsort+') for s_, ps, k, m in sets[:nsets]]
    sum((l * up.double()).sum() * w for l, w in zip(refs, (1.0, 0.5, 2.0))).backward()
    for got, ref in ((e.grad, e2.grad), (pr.grad, p2.grad)):
      scale = max(ref.abs().max().item(), 1e-9)
      assert (got.double() - ref).abs().max().item() <= 1e-5 * scale, (nsets, (got.double() - ref).abs().max().item(), scale)
    # per owner pixel: its gradient row against its own scale (six decades between rows)
    rows = e2.grad.abs().amax(1)
    err = (e.grad.double() - e2.grad).abs().amax(1)
    assert (err <= 2e-5 * rows + 1e-30).all(), float((err / rows.clamp_min(1e-30)).max())
    assert float(e.grad[up == 0].abs().max()) == 0.0
  # grouped single set == float64 with the other groups' prototypes masked out of every sum
  pg = T(np.sort((synth.hash_u64(201 + c, P) % np.uint64(3)).astype(np.int64)))
  qg = pg[inst]
  sem, psem = sets[0][0], sets[0][1]
  e, pr = T(e_np).requires_grad_(True), T(p_np).requires_grad_(True)
  nll = sl.segsort_nll(e, sem, inst, pr, psem, 12.0, 'segsort+', pixel_groups=qg, prototype_groups=pg)
  (nll * up).sum().backward()
  e2, p2 = T(e_np).double().requires_grad_(True), T(p_np).double().requires_grad_(True)
  sim = torch.exp(torch.mm(e2, p2.t()) * 12.0) * (qg.view(-1, 1) == pg.view(1, -1)).double()
  own = torch.gather(sim, 1, inst.view(-1, 1))
  same = (sem.view(-1, 1) == psem.view(1, -1)).double()
  sw = (sim * same).sum(1, keepdim=True) - own
  num = torch.where(sw > 0, sw, own)
  ref = -(num / ((sim * (1.0 - same)).sum(1, keepdim=True) + num)).log().view(-1)
  (ref * up.double()).sum().backward()
  for got, r in ((e.grad, e2.grad), (pr.grad, p2.grad)):
    scale = max(r.abs().max().item(), 1e-9)
    assert (got.double() - r).abs().max().item() <= 1e-5 * scale


def test_grouped_loss_equals_per_group_tables(dev, oracle):
  """pixel / prototype groups of the loss kernels (include/hsgk.h): the grouped call == one plain call per group
  on the compacted rows (forward per-pixel nll vs the oracle, both gradients vs the plain GPU calls); a pixel
  with zero upstream gradient whose own prototype lies outside its group contributes nothing."""
  import torch
  from hsg_amd.utils.segsort import loss as sl
  n, c, P, ngroups = 3000, 64, 90, 4
  e = oracle.normalize_embedding(synth.gaussish(71, n * c).reshape(n, c))
  pr = oracle.normalize_embedding(synth.gaussish(72, P * c).reshape(P, c))
  pg = np.sort((synth.hash_u64(73, P) % np.uint64(ngroups)).astype(np.int64))
  psem = (synth.hash_u64(74, P) % np.uint64(5)).astype(np.int64)
  inst = (synth.hash_u64(75, n) % np.uint64(P)).astype(np.int64)
  qg = pg[inst].copy()
  sem = psem[inst].copy()
  sem[::7] = (sem[::7] + 1) % 5
  T = lambda a: torch.from_numpy(a).to(dev)
  et, pt = T(e).requires_grad_(True), T(pr).requires_grad_(True)
  nll = sl.segsort_nll(et, T(sem), T(inst), pt, T(psem), 12.0, 'segsort+', pixel_groups=T(qg), prototype_groups=T(pg))
  w = T(synth.gaussish(76, n).astype(np.float32))
  (nll * w).sum().backward()
  want = np.zeros(n, np.float32)
  ge, gp = torch.zeros_like(et), torch.zeros_like(pt)
  for g in range(ngroups):
    pi, qi = np.nonzero(pg == g)[0], np.nonzero(qg == g)[0]
    remap = -np.ones(P, np.int64)
    remap[pi] = np.arange(len(pi))
    want[qi] = oracle.segsort_nll(e[qi], sem[qi], remap[inst[qi]], pr[pi], psem[pi], 12.0, 'segsort+').reshape(-1)
    e2, p2 = T(e[qi]).requires_grad_(True), T(pr[pi]).requires_grad_(True)
    part = sl.segsort_nll(e2, T(sem[qi]), T(remap[inst[qi]]), p2, T(psem[pi]), 12.0, 'segsort+')
    (part * w[T(qi)]).sum().backward()
    ge[T(qi)] += e2.grad
    gp[T(pi)] += p2.grad
  assert np.abs(nll.detach().cpu().numpy() - want).max() <= 1e-4
  assert (et.grad - ge).abs().max().item() <= 2e-5 * max(ge.abs().max().item(), 1.0)
  assert (pt.grad - gp).abs().max().item() <= 2e-5 * max(gp.abs().max().item(), 1.0)
  # own prototype outside the pixel's group: inf / nan forward, masked by the caller, no gradient at all
  qg2 = qg.copy()
  qg2[:50] = ngroups + 3
  e3 = T(e).requires_grad_(True)
  nll2 = sl.segsort_nll(e3, T(sem), T(inst), T(pr), T(psem), 12.0, 'segsort+', pixel_groups=T(qg2), prototype_groups=T(pg))
  keep = torch.ones(n, dtype=torch.bool, device=dev)
  keep[:50] = False
  torch.where(keep, nll2, torch.zeros_like(nll2)).sum().backward()
  assert torch.isfinite(e3.grad).all() and float(e3.grad[:50].abs().max()) == 0.0


@pytest.mark.parametrize('shape,grid,labelled', [((3, 256, 40, 56), (4, 4), True), ((2, 128, 65, 33), (2, 3), False)])
def test_pipelined_prep_variant_is_bit_identical(dev, oracle, monkeypatch, shape, grid, labelled):
  """HSGK_PREP=pipe: the persistent two-LDS-tile form of the prep kernel (kept for the A/B of DESIGN.md
  section 5b: it is slower) produces the same five outputs as the oracle."""
  monkeypatch.setenv('HSGK_PREP', 'pipe')
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + 321 + C, shape, 'mixture')
  lab = synth.overseg_labels(synth.SEED_BASE + 17, B, H, W, regions=5, ignore_rows=2) if labelled else None
  ign = 255 if labelled else None
  loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
  got = _run_segkm(dev, x, lab, grid, ign, 4)
  ref = oracle.segment_by_kmeans(x, lab, grid, loc, ign, 4)
  for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert a.shape == b.shape and np.array_equal(a, b), name


@pytest.mark.parametrize('switch', ['HSGK_PREP_FLAT=0', 'HSGK_PREP_ORDER=0', 'HSGK_PREP_X=6'])
@pytest.mark.parametrize('shape,grid,labelled', [((3, 256, 40, 56), (4, 4), True), ((2, 128, 65, 33), (2, 3), False),
                                                 ((1, 256, 64, 64), (16, 16), False)])
def test_prep_kernel_switches_are_bit_identical(dev, oracle, monkeypatch, switch, shape, grid, labelled):
  """The A/B switches of the round-6 prep kernel -- phase 3 as in rounds 2-5, workgroup ids as
  they come, the two scheduling experiments -- give the oracle's five outputs like the default does: with a label
  map + ignore band (compacted half tiles that start on odd rows: the 8-byte head / tail of the flat stream), an odd
  image size (ragged last tile) and a K = 256 shape (fp16 copy in tile order)."""
  name, val = switch.split('=')
  monkeypatch.setenv(name, val)
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + 654 + C, shape, 'mixture')
  lab = synth.overseg_labels(synth.SEED_BASE + 18, B, H, W, regions=5, ignore_rows=3) if labelled else None
  ign = 255 if labelled else None
  loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
  got = _run_segkm(dev, x, lab, grid, ign, 3)
  ref = oracle.segment_by_kmeans(x, lab, grid, loc, ign, 3)
  for nm, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert a.shape == b.shape and np.array_equal(a, b), nm


def _exchange_case(seed, sizes, C, nimg, ncl, nsem, ninst, shuffle_ids=False):
  """Per-source pixel sets with image-major rows (like segment_by_kmeans output) or arbitrary ids."""
  parts = []
  for g, n in enumerate(sizes):
    e = synth.gaussish(seed + 10 * g, n * C).reshape(n, C).astype(np.float64)
    e = (e / np.sqrt((e * e).sum(1, keepdims=True))).astype(np.float32)
    l = synth.gaussish(seed + 10 * g + 1, n * 2).reshape(n, 2) * np.float32(0.3)
    el = np.concatenate([e, l], 1).astype(np.float64)
    el = (el / np.sqrt((el * el).sum(1, keepdims=True))).astype(np.float32)
    img = np.sort((synth.hash_u64(seed + 10 * g + 2, n) % np.uint64(nimg)).astype(np.int64)) if n else np.zeros((0,), np.int64)
    cl = (synth.hash_u64(seed + 10 * g + 3, n) % np.uint64(ncl)).astype(np.int64)
    if shuffle_ids:
      cl = cl * 977 + 13                                   # sparse, large ids
    parts.append(dict(emb=e, emb_loc=el, cluster=cl, batch=img + (nimg * g if not shuffle_ids else 0),
                      sem=(synth.hash_u64(seed + 10 * g + 4, n) % np.uint64(nsem)).astype(np.int64),
                      inst=(synth.hash_u64(seed + 10 * g + 5, n) % np.uint64(ninst)).astype(np.int64)))
  return parts


@pytest.mark.parametrize('sizes,C,nimg,ncl,nsem,ninst,shuffle', [
    ([5000], 256, 3, 64, 1, 1, False),            # one source, rows of 256 / 258 columns, several chunks
    ([9000, 4100], 128, 2, 16, 3, 2, False),      # two sources, 130-column rows (scalar tail of 2)
    ([2500, 0, 3000], 16, 2, 6, 4, 3, False),     # an empty source; rows shorter than a slice
    ([6000, 6000], 64, 2, 40, 2, 2, True),        # the same images on both sources: segments MERGE; sparse ids
    ([4097], 384, 1, 128, 1, 1, False),           # 384 / 386 columns: three slices + tail
    ([3000], 200, 2, 9, 2, 1, False),             # 200 / 202 columns: remainder > 64 is its own slice
])
def test_exchange_list_mode_bit_exact_vs_oracle(dev, oracle, sizes, C, nimg, ncl, nsem, ninst, shuffle):
  """hsg/models/utils.py:127-217 through the list API (one process, several 'GPUs' -- all tensors on the one
  device of the box, each with its own backend / workspace): ids, labels exact and BOTH float tables
  bit-identical to oracle.exchange_prototypes (same C2 sums per source, same C1 norm chain)."""
  import torch
  from hsg_amd.models import utils as mu
  parts = _exchange_case(1234 + C, sizes, C, nimg, ncl, nsem, ninst, shuffle)
  T = lambda k: [torch.from_numpy(p[k]).to(dev) for p in parts]
  want = oracle.exchange_prototypes(parts)
  got = mu.gather_clustering_and_update_prototypes(T('emb'), T('emb_loc'), T('cluster'), T('batch'), T('sem'),
                                                   T('inst'), dev)
  for j, name in ((2, 'psem'), (3, 'pinst'), (4, 'pbatch')):
    assert np.array_equal(got[j][0].cpu().numpy(), want[j]), name
  for g in range(len(sizes)):
    assert np.array_equal(got[5][g].cpu().numpy(), want[5][g]), 'ids of source %d' % g
  assert np.array_equal(got[0][0].cpu().numpy().view(np.uint32), want[0].view(np.uint32)), 'prototypes'
  assert np.array_equal(got[1][-1].cpu().numpy().view(np.uint32), want[1].view(np.uint32)), 'prototypes_with_loc'
  # a single tensor (one process per GPU, world 1) takes the same kernels through _Exchange
  if len(sizes) == 1:
    one = mu.gather_clustering_and_update_prototypes(*[T(k)[0] for k in ('emb', 'emb_loc', 'cluster', 'batch', 'sem', 'inst')])
    assert np.array_equal(one[5].cpu().numpy(), want[5][0])
    assert np.array_equal(one[0].cpu().numpy().view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(one[1].cpu().numpy().view(np.uint32), want[1].view(np.uint32))


def test_exchange_capacity_regrowth_and_errors(dev, oracle):
  """The tuple blocks start too small (4 rows): the device reports the needed rows, the mirror regrows and
  repeats; negative values raise like the reference's scatter would."""
  import torch
  from hsg_amd.models import utils as mu
  parts = _exchange_case(77, [3000, 2000], 32, 2, 30, 2, 2)
  T = lambda k: [torch.from_numpy(p[k]).to(dev) for p in parts]
  want = oracle.exchange_prototypes(parts)
  saved = mu._CAP_START
  mu._capacity.clear()
  mu._CAP_START = 4
  try:
    got = mu.gather_clustering_and_update_prototypes(T('emb'), T('emb_loc'), T('cluster'), T('batch'), T('sem'),
                                                     T('inst'), dev)
    assert mu._cap_get(None, 'proto_list') >= max(int(np.unique(np.stack([p[k] for k in ('batch', 'cluster', 'sem', 'inst')], 1), axis=0).shape[0]) for p in parts)
  finally:
    mu._CAP_START = saved
    mu._capacity.clear()
  assert np.array_equal(got[5][1].cpu().numpy(), want[5][1])
  assert np.array_equal(got[0][0].cpu().numpy().view(np.uint32), want[0].view(np.uint32))
  bad = T('sem')
  bad[0][5] = -3
  with pytest.raises(ValueError):
    mu.gather_clustering_and_update_prototypes(T('emb'), T('emb_loc'), T('cluster'), T('batch'), bad, T('inst'), dev)


def test_exchange_gradients_vs_torch(dev):
  """backward through finish (normalise) and the row -> segment map, list mode, against torch autograd of the
  same formula."""
  import torch
  from hsg_amd.models import utils as mu
  parts = _exchange_case(99, [1500, 1100], 48, 2, 10, 2, 2)
  T = lambda k: [torch.from_numpy(p[k]).to(dev) for p in parts]
  embs = [t.requires_grad_(True) for t in T('emb')]
  locs = [t.requires_grad_(True) for t in T('emb_loc')]
  got = mu.gather_clustering_and_update_prototypes(embs, locs, T('cluster'), T('batch'), T('sem'), T('inst'), dev)
  P = got[0][0].shape[0]
  w1 = torch.from_numpy(synth.gaussish(5, P * 48).reshape(P, 48)).to(dev)
  w2 = torch.from_numpy(synth.gaussish(6, P * 50).reshape(P, 50)).to(dev)
  ((got[0][0] * w1).sum() + (got[1][1] * w2).sum()).backward()
  e2 = [t.detach().clone().requires_grad_(True) for t in embs]
  l2 = [t.detach().clone().requires_grad_(True) for t in locs]
  ids = torch.cat(got[5])
  sa = torch.zeros((P, 48), device=dev).index_add(0, ids, torch.cat(e2))
  sb = torch.zeros((P, 50), device=dev).index_add(0, ids, torch.cat(l2))
  pa = sa / sa.norm(dim=1, keepdim=True).clamp_min(1e-12)
  pb = sb / sb.norm(dim=1, keepdim=True).clamp_min(1e-12)
  ((pa * w1).sum() + (pb * w2).sum()).backward()
  for a, b in zip(embs + locs, e2 + l2):
    assert (a.grad - b.grad).abs().max().item() <= 2e-5 * max(b.grad.abs().max().item(), 1.0)


def test_exchange_c_abi_composite_one_call(dev, oracle):
  """hsgk_exchange_prototypes (begin + finish over the whole table capacity, no host read in between) through
  ctypes, world 1: without a communicator, and with a ONE-rank RCCL communicator made by hsgk_comm_*; a
  starved partial-row pool (chunks fall back to the per-segment scan) gives the same bits."""
  import ctypes
  import torch
  from hsg_amd import _lib
  parts = _exchange_case(4321, [7000], 128, 3, 20, 2, 2)
  want = oracle.exchange_prototypes(parts)
  p = parts[0]
  L = _lib.lib()
  n, C, D = p['emb'].shape[0], 128, 130
  P = want[0].shape[0]
  t = {k: torch.from_numpy(v).to(dev) for k, v in p.items()}
  comm = ctypes.c_void_p()
  ident = (ctypes.c_uint8 * 128)()
  _lib.check(L.hsgk_comm_unique_id(ident, 128))
  _lib.check(L.hsgk_comm_init_rank(ctypes.byref(comm), 1, 0, ident, 128))
  try:
    for use_comm, pool_rows in ((False, 4096), (True, 4096), (False, 40)):
      cap = 512
      wsb = L.hsgk_exchange_workspace_bytes(n, C, D, cap, cap, 1, pool_rows)
      ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
      table = torch.empty((cap, C + D), device=dev)
      pa, pb = torch.empty((cap, C), device=dev), torch.empty((cap, D), device=dev)
      norms = torch.empty((cap, 2), device=dev)
      plab = torch.empty((3, cap), dtype=torch.int64, device=dev)
      upd = torch.empty((n,), dtype=torch.int64, device=dev)
      meta = torch.empty((8,), dtype=torch.int64, device=dev)
      a = _lib.ExchangeArgs(
          embeddings=t['emb'].data_ptr(), embeddings_loc=t['emb_loc'].data_ptr(), cluster=t['cluster'].data_ptr(),
          batch=t['batch'].data_ptr(), semantic=t['sem'].data_ptr(), instance=t['inst'].data_ptr(), n=n, C=C, D=D,
          cap_local=cap, cap_total=cap, pool_rows=pool_rows, eps=_lib.EPS, table=table.data_ptr(),
          prototypes=pa.data_ptr(), prototypes_loc=pb.data_ptr(), norms=norms.data_ptr(),
          proto_semantic=plab[0].data_ptr(), proto_instance=plab[1].data_ptr(), proto_batch=plab[2].data_ptr(),
          updated_cluster=upd.data_ptr(), meta=meta.data_ptr(), workspace=ws.data_ptr(), workspace_bytes=wsb)
      _lib.check(L.hsgk_exchange_prototypes(ctypes.byref(a), comm if use_comm else None, 0, 1, _lib.stream_ptr()))
      m = meta.cpu().tolist()
      assert m[0] == P and m[1] == P and m[2] == 0 and m[3] == P, m
      assert np.array_equal(upd.cpu().numpy(), want[5][0])
      assert np.array_equal(plab[0, :P].cpu().numpy(), want[2]) and np.array_equal(plab[2, :P].cpu().numpy(), want[4])
      assert np.array_equal(pa[:P].cpu().numpy().view(np.uint32), want[0].view(np.uint32)), (use_comm, pool_rows)
      assert np.array_equal(pb[:P].cpu().numpy().view(np.uint32), want[1].view(np.uint32)), (use_comm, pool_rows)
      assert float(pa[P:].abs().max()) == 0.0          # unused table rows: zero sums -> zero rows
    # the in-place all_reduce helper on the one-rank communicator is the identity
    buf = torch.arange(1000, dtype=torch.float32, device=dev)
    _lib.check(L.hsgk_comm_all_reduce_f32(buf.data_ptr(), 1000, comm, _lib.stream_ptr()))
    assert torch.equal(buf.cpu(), torch.arange(1000, dtype=torch.float32))
  finally:
    _lib.check(L.hsgk_comm_destroy(comm))


@pytest.mark.parametrize('B,HW,C,K', [(2, 4096, 32, 8), (1, 5000, 256, 64), (3, 2500, 128, 37),
                                      (1, 300, 64, 64), (2, 3000, 384, 128), (1, 4000, 256, 100), (1, 3000, 256, 200),
                                      (2, 2500, 256, 256)])
def test_split_estep_matches_exact_labels(dev, oracle, B, HW, C, K):
  """Filtered E-steps == canonical fp32 argmax: unit_rows = 2 (fp16 copy ->
  bf16x3 on the undecided rows -> exact chains), 1 (bf16x3 -> exact) and 0 (pure
  fp32 kernel), including exact ties (duplicate centroids), near ties at the
  scale of each filter's gap (perturbed copies) and zero centroids.  64 < K <= 128 takes
  the hi-plane fp16 filter straight to the exact chains, 128 < K <= 256 the same in two
  table halves (unit_rows = 2)."""
  import torch
  from hsg_amd import _lib
  D = C + 2
  n = B * HW
  x = oracle.normalize_embedding(synth.gaussish(31 + C, n * D).reshape(n, D))
  cent = oracle.normalize_embedding(synth.gaussish(37 + K, B * K * D).reshape(B * K, D))
  cent = cent.reshape(B, K, D).copy()
  if K >= 8:
    cent[:, 3] = cent[:, 1]                                   # exact tie -> first index
    near = cent[:, 2] + np.float32(3e-6) * cent[:, 5]
    cent[:, 6] = oracle.normalize_embedding(near)             # gap ~1e-6: must be re-scored exactly
    mid = cent[:, 4] + np.float32(4e-4) * cent[:, 7]
    cent[:, 0] = oracle.normalize_embedding(mid)              # gap ~1e-4: beyond the fp16 filter only
    cent[:, K - 1] = 0.0                                      # empty cluster
  if K >= 24:
    # a cloud of near copies of one centroid (gaps ~1e-5: inside every filter's gap): rows close to
    # it carry 4..7 candidates in the exact queue, more than seven when the second table half joins in
    for i, kk in enumerate((9, 10, 11, 12, 13)):
      cent[:, kk] = oracle.normalize_embedding(cent[:, 8] + np.float32(1e-5 * (i + 1)) * cent[:, 14 + i])
    if K > 140:
      for i, kk in enumerate((130, 131, 133, 139)):
        cent[:, kk] = oracle.normalize_embedding(cent[:, 8] + np.float32(1.5e-5 * (i + 1)) * cent[:, 20 + i])
    # and rows that sit on that centroid
    x[::17] = oracle.normalize_embedding(cent[0, 8][None, :] + np.float32(0.02) * x[::17])
  L = _lib.lib()
  xt = torch.from_numpy(x).to(dev)
  ct = torch.from_numpy(cent).to(dev)
  wsb = L.hsgk_lloyd_workspace_bytes(B, HW, D, K)
  ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
  got = {}
  for unit in (2, 1, 0):
    out = torch.full((n,), -1, dtype=torch.int32, device=dev)
    _lib.check(L.hsgk_lloyd_estep(xt.data_ptr(), B, HW, D, K, ct.data_ptr(), out.data_ptr(), unit,
                                  ws.data_ptr(), wsb, _lib.stream_ptr()))
    got[unit] = out.cpu().numpy().astype(np.int64)
  for b in range(B):
    ref = oracle.find_nearest_prototypes(x[b * HW:(b + 1) * HW], cent[b])
    assert np.array_equal(got[0][b * HW:(b + 1) * HW], ref), 'fp32 kernel'
    assert np.array_equal(got[1][b * HW:(b + 1) * HW], ref), 'split kernel'
    assert np.array_equal(got[2][b * HW:(b + 1) * HW], ref), 'fp16 filter first'


@pytest.mark.parametrize('fixture', ['f7_hierarchy', 'f7_hierarchy_multiview', 'f7_hierarchy_multiview_b',
                                     'f7_hierarchy_m256', 'f7_hierarchy_div2048'])
def test_hierarchy_ops_vs_reference_golden(dev, fixture):
  """a10-a14: padded per-image prototypes (base and MULTIVIEW variant: the views of one image
  stacked in one row, `image_indices`), softmax/argmax/Bayes-chain grouping, group means and the
  pixel label lookup against the reference's own methods (tests/golden/f7_*.npz, generated by
  calling them with stub selves); `_m256` = BASELINE.json configs[3]'s 256 -> 64 -> 16 sizes."""
  import torch
  from hsg_amd.models.embeddings import hierarchy as hz
  g = util.load(fixture)
  M, KF, KC = int(g['M']), int(g['KF']), int(g['KC'])
  div = int(g['label_divisor']) if 'label_divisor' in g else 256     # `_div2048`: 16 views, panoptic-coded labels
  seed = int(g['seed'])
  _, C, H, W = (int(v) for v in g['shape'])
  emb = torch.from_numpy(g['emb']).to(dev).requires_grad_(True)
  n = emb.shape[0]
  pos = torch.from_numpy(synth.gaussish(seed + 1, n * C).reshape(n, C).copy()).to(dev)
  T = lambda k: torch.from_numpy(g[k]).to(dev)
  img_idx = T('image_indices') if g['image_indices'].size else None
  protos, pos_protos, masks, plabs, pbatch, c_by_img = hz.calculate_kmeans_prototypes(
      emb, T('cidx'), T('bidx'), pos, T('labels'), img_idx, label_divisor=div, max_num_clusters=M)
  B = protos.shape[0]
  assert np.array_equal(masks.cpu().numpy(), g['masks'])
  assert np.array_equal(plabs.cpu().numpy(), g['plabs'])
  assert np.array_equal(pbatch.cpu().numpy(), g['pbatch'])
  assert np.array_equal(c_by_img.cpu().numpy(), g['c_by_img'])
  assert np.abs(protos.detach().cpu().numpy() - g['protos']).max() <= FTOL
  assert np.abs(pos_protos.cpu().numpy() - g['pos_protos']).max() <= 1e-5
  protos.sum().backward()
  assert emb.grad is not None and torch.isfinite(emb.grad).all()
  # the Cityscapes twin (resnet_fcn_hsg_cs.py:499-502, :1061-1064): the same tables, padded only to the largest
  # number of clusters of an image
  dyn = hz.calculate_kmeans_prototypes(emb.detach(), T('cidx'), T('bidx'), pos, T('labels'), img_idx,
                                       label_divisor=div, max_num_clusters=None)
  Md = int((~masks).sum(1).max())
  assert dyn[0].shape[2] == Md and dyn[2].shape[1] == Md
  assert torch.equal(dyn[0], protos.detach()[:, :, :Md]) and torch.equal(dyn[1], pos_protos[:, :, :Md])
  for a, b2 in zip(dyn[2:5], (masks, plabs, pbatch)):
    assert torch.equal(a, b2[:, :Md])
  assert torch.equal(dyn[5], c_by_img)

  fine_logits = (torch.from_numpy(synth.gaussish(seed + 2, B * KF * M).reshape(B, KF, M).copy()) * 2).to(dev)
  coarse_logits = (torch.from_numpy(synth.gaussish(seed + 3, B * KC * KF).reshape(B, KC, KF).copy()) * 2).to(dev)
  fl = fine_logits.clone().requires_grad_(True)
  cl = coarse_logits.clone().requires_grad_(True)
  f_lab, f_prob, c_lab, c_prob = hz.hierarchical_grouping_from_logits(fl, cl)
  assert np.array_equal(f_lab.cpu().numpy(), g['f_lab'])
  assert np.array_equal(c_lab.cpu().numpy(), g['c_lab'])
  fp = f_prob.detach().cpu().numpy()
  assert np.abs((fp if g['f_prob'].shape == fp.shape else fp[:, ::7]) - g['f_prob']).max() <= 1e-6
  assert np.abs(c_prob.detach().cpu().numpy() - g['c_prob']).max() <= 1e-6
  # gradients of the fused op == gradients of the ATen formulation
  w = torch.from_numpy(synth.gaussish(seed + 9, B * KC * M).reshape(B, KC, M).copy()).to(dev)
  (c_prob * w).sum().backward()
  a = fine_logits.clone().requires_grad_(True)
  b2 = coarse_logits.clone().requires_grad_(True)
  ref = torch.einsum('bij,bjk->bik', torch.softmax(b2, 1), torch.softmax(a, 1))
  (ref * w).sum().backward()
  assert (fl.grad - a.grad).abs().max().item() <= 1e-6
  assert (cl.grad - b2.grad).abs().max().item() <= 1e-6

  fine_pos = hz.collect_nd_coarser_prototype(T('pos_protos'), T('f_lab'), T('masks'), KF, False)
  fine_pos_n = hz.collect_nd_coarser_prototype(T('protos'), T('f_lab'), T('masks'), KF, True)
  assert np.abs(fine_pos.cpu().numpy() - g['fine_pos']).max() <= 1e-5
  assert np.abs(fine_pos_n.cpu().numpy() - g['fine_pos_n']).max() <= 2e-6
  pp = T('protos').requires_grad_(True)
  hz.collect_nd_coarser_prototype(pp, T('f_lab'), T('masks'), KF, True).sum().backward()
  assert torch.isfinite(pp.grad).all()

  # generate_clusters:942-957: the lookups are keyed by image id in the multiview model
  px_ids = T('bidx') if img_idx is None else img_idx[T('bidx')]
  px_fine = hz.collect_pixel_hierarchical_clustering_indices(T('c_by_img'), px_ids, T('f_lab'))
  px_coarse = hz.collect_pixel_hierarchical_clustering_indices(T('c_by_img'), px_ids, T('c_lab'))
  assert np.array_equal(px_fine.cpu().numpy(), g['px_fine'])
  assert np.array_equal(px_coarse.cpu().numpy(), g['px_coarse'])


@pytest.mark.parametrize('B,C,N,G,normalized,masked', [(4, 128, 256, 8, True, True), (2, 256, 256, 64, True, True),
                                                       (3, 70, 33, 5, False, True), (1, 32, 100, 4, True, False)])
@pytest.mark.parametrize('binding', ['torch', 'ctypes'])
def test_group_mean_backward_kernel_vs_autograd(dev, monkeypatch, binding, B, C, N, G, normalized, masked):
  """hsgk_group_mean_bwd (round 6: one launch instead of autograd through the ATen restatement of
  resnet_fcn_hsg.py:706-746) against float64 autograd of that restatement: empty groups, padded nodes, a group
  whose mean is (numerically) zero takes the clamped branch; through both host bindings."""
  import torch
  from hsg_amd.models.embeddings import hierarchy as hz
  if binding == 'ctypes':
    monkeypatch.setenv('HSGK_BINDING', 'ctypes')
  g = torch.Generator(device=dev).manual_seed(B * 1000 + C + N + G)
  p = torch.randn((B, C, N), device=dev, generator=g)
  lab = torch.randint(0, G, (B, N), device=dev, generator=g)
  lab[lab == G - 1] = 0                                   # group G - 1 stays empty
  masks = (torch.rand((B, N), device=dev, generator=g) < 0.2) if masked else None
  if N >= 8:                                              # a group of two nodes that cancel: mean 0 -> clamped norm
    lab[0, :2] = 1
    lab[0, 2:][lab[0, 2:] == 1] = 0
    p[0, :, 1] = -p[0, :, 0]
    if masks is not None:
      masks[0, :2] = False
  w = torch.randn((B, C, G), device=dev, generator=g)
  a = p.clone().requires_grad_(True)
  out = hz.collect_nd_coarser_prototype(a, lab, masks, G, normalized)
  (out * w).sum().backward()
  b2 = p.double().requires_grad_(True)
  ref = hz._group_mean_torch(b2, lab, masks, G, normalized)
  (ref * w.double()).sum().backward()
  assert (out.double() - ref).abs().max().item() <= 2e-6
  want = b2.grad
  # (the clamped group's gradient is g / eps = 1e12 x g in both; compare relative to each entry's own scale)
  err = (a.grad.double() - want).abs() / want.abs().clamp(min=1.0)
  assert err.max().item() <= 2e-5, err.max().item()


@pytest.mark.parametrize('B,KF,KC,N', [(3, 8, 4, 256), (2, 44, 9, 700), (1, 5, 0, 33), (4, 16, 2, 64)])
def test_hier_assign_backward_kernel_vs_autograd(dev, B, KF, KC, N):
  """hsgk_hier_assign_bwd (softmax backward of both levels and the chain through coarse_prob = softmax(coarse) x
  fine_prob in one launch) against float64 autograd of the ATen formulation, with weights on both probability
  outputs, on one of them only, and without a coarse level."""
  import torch
  from hsg_amd.models.embeddings import hierarchy as hz
  g = torch.Generator(device=dev).manual_seed(B * 100 + KF + N)
  fine = 2 * torch.randn((B, KF, N), device=dev, generator=g)
  coarse = 2 * torch.randn((B, KC, KF), device=dev, generator=g) if KC else None
  w1 = torch.randn((B, KF, N), device=dev, generator=g)
  w2 = torch.randn((B, max(KC, 1), N), device=dev, generator=g)
  for use1, use2 in ((True, True), (False, True), (True, False)):
    if not KC and not use1:
      continue
    fl = fine.clone().requires_grad_(True)
    cl = coarse.clone().requires_grad_(True) if KC else None
    out = hz.hierarchical_grouping_from_logits(fl, cl)
    f_prob = out[1]
    loss = (f_prob * w1).sum() * (1.0 if use1 else 0.0)
    if KC and use2:
      loss = loss + (out[3] * w2).sum()
    if not use1 and not (KC and use2):
      continue
    loss.backward()
    a = fine.double().requires_grad_(True)
    b2 = coarse.double().requires_grad_(True) if KC else None
    pf = torch.softmax(a, 1)
    ref = (pf * w1.double()).sum() * (1.0 if use1 else 0.0)
    if KC and use2:
      ref = ref + (torch.einsum('bij,bjk->bik', torch.softmax(b2, 1), pf) * w2.double()).sum()
    ref.backward()
    assert (fl.grad.double() - a.grad).abs().max().item() <= 2e-6 * max(a.grad.abs().max().item(), 1.0)
    if KC:
      want = b2.grad if b2.grad is not None else torch.zeros_like(b2)
      got = cl.grad if cl.grad is not None else torch.zeros_like(cl)
      assert (got.double() - want).abs().max().item() <= 2e-5 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize('shape,grid,iters', [
    ((2, 64, 64, 64), (16, 16), 5),      # cfg4-style K = 256: four 64-row table blocks, K-blocked M-step
    ((1, 384, 32, 48), (8, 16), 6),      # cfg5-style C = 384, K = 128
    ((3, 256, 40, 56), (8, 8), 10),      # cfg2/3-style C = 256, K = 64: split E-step, ragged chunks
    ((2, 128, 33, 47), (5, 7), 7),       # odd sizes, K = 35, d = 130 (split shape ok: 4 chunks)
    ((1, 512, 24, 40), (4, 5), 4),       # C = 512: two plane-load batches in prep, two 16-byte vectors per lane in the sums update
    ((2, 64, 50, 30), (5, 3), 3),        # C = 64: quarter-filled waves in prep, d = 66 (no fp16 level)
    ((1, 320, 20, 36), (2, 3), 5),       # C = 320 = 5 x 64: fp16 level with an odd number of 64-column steps
])
def test_baseline_config_shapes_vs_oracle(dev, oracle, shape, grid, iters):
  """Reduced-size versions of BASELINE.json configs 2-5, bit-exact vs the oracle
  (labels, ids and both float outputs)."""
  from hsg_amd.utils.segsort import common as sc
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + C + H, shape, 'iid')
  lab = synth.overseg_labels(synth.SEED_BASE + 3, B, H, W, regions=7, ignore_rows=3)
  loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
  got = _run_segkm(dev, x, lab, grid, 255, iters)
  ref = oracle.segment_by_kmeans(x, lab, grid, loc, 255, iters)
  for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert a.shape == b.shape, name
    assert np.array_equal(a, b), '%s: %d mismatching elements' % (name, int((a != b).sum()))


@pytest.mark.parametrize('small', ['1', '0'])
@pytest.mark.parametrize('shape,grid,iters,flavour', [
    ((5, 256, 28, 28), (8, 8), 10, 'iid'),        # the reference's training resolution (448 / 16), K = 64
    ((3, 256, 14, 14), (8, 8), 7, 'mixture'),     # 224 / 16: clusters empty out
    ((2, 128, 20, 31), (4, 5), 5, 'iid'),         # C = 128 (d = 130: prefetch depth 2), ragged
    ((4, 256, 24, 16), (6, 6), 1, 'mixture'),     # a single iteration (labels end in the other buffer)
    ((2, 256, 32, 32), (8, 8), 4, 'iid'),         # 1024 rows: the largest image the fused kernel takes
])
def test_small_maps_fused_and_per_kernel_routes_vs_oracle(dev, oracle, monkeypatch, shape, grid, iters, flavour, small):
  """Training-resolution feature maps: the whole Lloyd loop of an image in one workgroup
  (lloyd_small_kernel, HSGK_SMALL=1, the default below 1024 rows per image) and the per-kernel route
  (HSGK_SMALL=0) -- both bit-exact vs the oracle, with labels + ignore band and without labels."""
  monkeypatch.setenv('HSGK_SMALL', small)
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + 3 * C + H, shape, flavour)
  loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
  for lab, ign in ((synth.overseg_labels(synth.SEED_BASE + 8, B, H, W, regions=4, ignore_rows=2), 255), (None, None)):
    got = _run_segkm(dev, x, lab, grid, ign, iters)
    ref = oracle.segment_by_kmeans(x, lab, grid, loc, ign, iters)
    for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
      assert a.shape == b.shape, name
      assert np.array_equal(a, b), '%s: %d mismatching elements' % (name, int((a != b).sum()))


@pytest.mark.parametrize('shape,grid,iters,m0,groups', [
    ((2, 128, 56, 56), (4, 4), 15, '0', '0'),     # the reference's own training hyper-parameters (bashscripts/*/train.sh)
    ((2, 128, 56, 56), (4, 4), 4, '0', '5'),
    ((5, 256, 28, 28), (8, 8), 10, '0', '0'),
    ((5, 256, 28, 28), (8, 8), 3, '0', '3'),
    ((2, 256, 40, 50), (8, 8), 6, '0', '8'),      # 2000 rows per image, labels + ignore: ragged shares
    ((3, 256, 24, 30), (2, 3), 3, '1', '2'),      # sums of the first M-step come from the prep kernel
])
def test_small_maps_several_workgroups_per_image_vs_oracle(dev, oracle, monkeypatch, shape, grid, iters, m0, groups):
  """The fused Lloyd kernel with several co-operating workgroups per image (each owns a share of the rows;
  the running sums are exchanged through device-scope atomics and a per-image tick counter):
  HSGK_SMALL_GROUPS forces the number of workgroups per image ('0': the library's choice)."""
  monkeypatch.setenv('HSGK_SMALL', '1')
  monkeypatch.setenv('HSGK_M0', m0)
  if groups != '0':
    monkeypatch.setenv('HSGK_SMALL_GROUPS', groups)
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + 5 * C + W, shape, 'iid')
  loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
  for lab, ign in ((synth.overseg_labels(synth.SEED_BASE + 11, B, H, W, regions=4, ignore_rows=3), 255), (None, None)):
    got = _run_segkm(dev, x, lab, grid, ign, iters)
    ref = oracle.segment_by_kmeans(x, lab, grid, loc, ign, iters)
    for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
      assert a.shape == b.shape, name
      assert np.array_equal(a, b), '%s: %d mismatching elements' % (name, int((a != b).sum()))


def test_small_maps_two_streams_concurrently(dev):
  """Two streams issuing multi-workgroup small-map calls at the same time: the co-residency cap keeps both
  grids within the CUs (no timeout, no deadlock) and every result equals the single-stream one.  (A second
  kernel competing for the caches is what exposed stale centroids when all workgroups of an image wrote ONE
  global copy from different XCDs; tools/probes/two_stream_small.py runs thousands of pairs.)"""
  import torch
  from hsg_amd.utils.segsort import common as sc
  shape, grid, iters = (20, 128, 40, 40), [4, 4], 8
  g = torch.Generator(device=dev).manual_seed(5)
  xs = [torch.randn(shape, device=dev, generator=g) for _ in range(2)]
  ref = [[t.clone() for t in sc.segment_by_kmeans(x, None, grid, iterations=iters)] for x in xs]
  torch.cuda.synchronize()
  streams = [torch.cuda.Stream() for _ in range(2)]
  for _ in range(150):
    outs = []
    for i, st in enumerate(streams):
      with torch.cuda.stream(st):
        outs.append(sc.segment_by_kmeans(xs[i], None, grid, iterations=iters))
    torch.cuda.synchronize()
    for o, r in zip(outs, ref):
      for a, b in zip(o, r):
        assert torch.equal(a, b)


def test_small_maps_beside_a_saturating_stream(dev):
  """The multi-workgroup small-map route (plain launch within half the device, at most two in flight, wall-clock bounded waits) while a SECOND stream
  keeps every CU busy with long streaming kernels (a backbone / collective stand-in): the reference's own
  training shape, with and without labels -- no error 3, no hang, every result bit-identical to the quiet run."""
  import torch
  from hsg_amd import _lib
  from hsg_amd.utils.segsort import common as sc
  shape, grid, iters = (4, 128, 56, 56), [4, 4], 15
  assert _lib.lib().hsgk_small_map_groups(4, 128, 56, 56, 16) > 1
  g = torch.Generator(device=dev).manual_seed(11)
  x = torch.randn(shape, device=dev, generator=g)
  lab = torch.from_numpy(synth.overseg_labels(synth.SEED_BASE + 3, 4, 56, 56, regions=5, ignore_rows=2)).to(dev)
  ref = [[t.clone() for t in sc.segment_by_kmeans(x, None, grid, iterations=iters)],
         [t.clone() for t in sc.segment_by_kmeans(x, lab, grid, ignore_index=255, iterations=iters)]]
  torch.cuda.synchronize()
  big = torch.empty((1 << 30,), dtype=torch.float32, device=dev)            # 4 GiB: ~2.5 ms per pass, all CUs
  side = torch.cuda.Stream()
  for rnd in range(6):
    with torch.cuda.stream(side):
      for _ in range(40):
        big.mul_(1.0001)
    outs = []
    for _ in range(10):
      outs.append(sc.segment_by_kmeans(x, None, grid, iterations=iters))
      outs.append(sc.segment_by_kmeans(x, lab, grid, ignore_index=255, iterations=iters))
    torch.cuda.synchronize()
    _lib.poll_deferred(wait=True)                                               # a timed-out wait would raise here
    for i, o in enumerate(outs):
      for a, b in zip(o, ref[i % 2]):
        assert torch.equal(a, b)


def test_boundary_is_thread_safe_one_thread_per_stream(dev, oracle):
  """The reference calls the operators from ONE PYTHON THREAD PER GPU (lib/nn/parallel/data_parallel.py:104-105).
  Two threads, each on its own stream (its own device when the box has two), run segment_by_kmeans, the
  prototype exchange, segment_reduce and the loss concurrently: every result equals the serial one."""
  import threading
  import torch
  from hsg_amd.models import utils as mu
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort.loss import SegSortLoss
  ndev = min(2, torch.cuda.device_count())
  devs = [torch.device('cuda', i % ndev) for i in range(2)]
  shapes = [((3, 128, 40, 56), [2, 4]), ((2, 256, 33, 47), [3, 3])]

  def work(i, reps, out):
    d = devs[i]
    torch.cuda.set_device(d)
    st = torch.cuda.Stream(device=d)
    shape, grid = shapes[i]
    x = torch.from_numpy(synth.embeddings_nchw(synth.SEED_BASE + 40 + i, shape, 'mixture')).to(d)
    lab = torch.from_numpy(synth.overseg_labels(synth.SEED_BASE + 50 + i, shape[0], shape[2], shape[3],
                                                regions=5, ignore_rows=2)).to(d)
    res = []
    with torch.cuda.stream(st):
      for _ in range(reps):
        emb, eloc, labels, cidx, bidx = sc.segment_by_kmeans(x, lab, grid, ignore_index=255, iterations=6)
        protos, protos_loc, psem, pinst, pbatch, upd = mu.gather_clustering_and_update_prototypes(
            emb, eloc, cidx, bidx, labels, torch.zeros_like(labels))
        means = sc.calculate_prototypes_from_labels(emb, upd, protos.shape[0])
        loss = SegSortLoss(16, 'segsort+')(emb, labels, upd, protos, psem)
        res.append([t.detach().cpu() for t in (emb, eloc, labels, cidx, protos, protos_loc, upd, means, loss)])
      st.synchronize()
    out[i] = res

  serial = [None, None]
  for i in range(2):
    work(i, 1, serial)
  conc = [None, None]
  threads = [threading.Thread(target=work, args=(i, 12, conc)) for i in range(2)]
  for t in threads:
    t.start()
  for t in threads:
    t.join()
  for i in range(2):
    assert conc[i] is not None and len(conc[i]) == 12, 'thread %d died' % i
    for rep in conc[i]:
      for a, b in zip(rep, serial[i][0]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('iters', [1, 2, 5])
def test_small_maps_fused_route_with_first_mstep_from_prep(dev, oracle, monkeypatch, iters):
  """The fused per-image Lloyd kernel starting from the sums the PREP kernel left for the seed labels
  (HSGK_M0=1 forces that fusion also for narrow seed cells): its first iteration then skips the M-step."""
  monkeypatch.setenv('HSGK_M0', '1')
  monkeypatch.setenv('HSGK_SMALL', '1')
  shape, grid = (3, 256, 24, 30), (2, 3)
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + 123, shape, 'iid')
  lab = synth.overseg_labels(synth.SEED_BASE + 9, B, H, W, regions=3, ignore_rows=1)
  loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
  got = _run_segkm(dev, x, lab, grid, 255, iters)
  ref = oracle.segment_by_kmeans(x, lab, grid, loc, 255, iters)
  for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert np.array_equal(a, b), '%s: %d mismatching elements' % (name, int((a != b).sum()))


@pytest.mark.parametrize('m0', ['1', '0'])
@pytest.mark.parametrize('shape,grid,iters', [((3, 256, 40, 56), (8, 8), 5), ((2, 128, 64, 96), (2, 3), 6),
                                              ((2, 64, 33, 47), (5, 7), 3)])
def test_first_mstep_fused_or_not_vs_oracle(dev, oracle, monkeypatch, shape, grid, iters, m0):
  """The first M-step inside the prep kernel (HSGK_M0=1: forced also for narrow seed cells, where pixels of a
  third label go through global atomics) or through the update kernel (HSGK_M0=0) -- both bit-exact."""
  monkeypatch.setenv('HSGK_M0', m0)
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + 7 * C + W, shape, 'iid')
  lab = synth.overseg_labels(synth.SEED_BASE + 6, B, H, W, regions=5, ignore_rows=3)
  loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
  got = _run_segkm(dev, x, lab, grid, 255, iters)
  ref = oracle.segment_by_kmeans(x, lab, grid, loc, 255, iters)
  for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert a.shape == b.shape, name
    assert np.array_equal(a, b), '%s: %d mismatching elements' % (name, int((a != b).sum()))


@pytest.mark.parametrize('level2', ['1', '0'])
@pytest.mark.parametrize('shape,grid,iters', [((3, 256, 40, 56), (8, 8), 6), ((2, 128, 33, 47), (5, 7), 4),
                                              ((1, 256, 90, 70), (4, 6), 8)])
def test_both_estep_routes_for_small_tables_vs_oracle(dev, oracle, monkeypatch, shape, grid, iters, level2):
  """K <= 64: the undecided rows of the fp16 level go through the bf16x3 level (HSGK_L2=1: the route of
  large batches) or straight to the exact chains (HSGK_L2=0: the default below 1.5 M rows) -- both bit-exact."""
  from hsg_amd.utils.segsort import common as sc
  monkeypatch.setenv('HSGK_L2', level2)
  B, C, H, W = shape
  x = synth.embeddings_nchw(synth.SEED_BASE + 5 * C + H, shape, 'mixture')
  lab = synth.overseg_labels(synth.SEED_BASE + 4, B, H, W, regions=6, ignore_rows=2)
  loc = oracle.generate_location_features((H, W)) - np.float32(0.5)
  got = _run_segkm(dev, x, lab, grid, 255, iters)
  ref = oracle.segment_by_kmeans(x, lab, grid, loc, 255, iters)
  for name, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert a.shape == b.shape, name
    assert np.array_equal(a, b), '%s: %d mismatching elements' % (name, int((a != b).sum()))


def test_full_size_cfg2_properties_and_spot_parity(dev, oracle):
  """BASELINE.json configs[1] at FULL size (48x256x448x448, K=8x8, 10 iterations):
  size-independent properties on the whole batch and bit-exact parity of three
  whole images against the oracle (all 48: tests/checkers/full_parity_cfg2.py)."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  B, C, H, W, grid, iters = 48, 256, 448, 448, (8, 8), 10
  gen = torch.Generator(device=dev)
  gen.manual_seed(1234)
  x = torch.randn((B, C, H, W), device=dev, generator=gen)
  emb, eloc, labels, cluster, batch = sc.segment_by_kmeans(x, None, list(grid), iterations=iters)
  n = B * H * W
  assert emb.shape == (n, C) and eloc.shape == (n, C + 2)
  # unit rows
  assert (emb.norm(dim=1) - 1).abs().max().item() < 1e-5
  assert (eloc.norm(dim=1) - 1).abs().max().item() < 1e-5
  # bookkeeping: image-major batch index, dense ids, ids sorted by (image, cluster)
  assert torch.equal(batch, torch.arange(B, device=dev).repeat_interleave(H * W))
  assert int(cluster.min()) == 0 and int(cluster.max()) == B * 64 - 1
  assert torch.equal(cluster // 64, batch)
  assert torch.equal(labels, torch.zeros_like(labels))
  # every cluster of the 448x448 grid survives on i.i.d. data
  assert int(torch.bincount(cluster).min()) > 0
  # determinism: a second run is bit-identical
  emb2, eloc2, _, cluster2, _ = sc.segment_by_kmeans(x, None, list(grid), iterations=iters)
  assert torch.equal(cluster, cluster2) and torch.equal(eloc, eloc2) and torch.equal(emb, emb2)
  del emb2, eloc2, cluster2
  # Lloyd fixed-point property of the final labels: re-assigning with the
  # centroids of the last M-step reproduces them (E-step idempotence), checked
  # through the exact fp32 kernel on one image
  loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
  for b in (0, 23, B - 1):
    ref = oracle.segment_by_kmeans(x[b:b + 1].cpu().numpy(), None, grid, loc, None, iters)
    sl = slice(b * H * W, (b + 1) * H * W)
    assert np.array_equal(emb[sl].cpu().numpy(), ref[0]), 'emb image %d' % b
    assert np.array_equal(eloc[sl].cpu().numpy(), ref[1]), 'emb_loc image %d' % b
    assert np.array_equal((cluster[sl] - b * 64).cpu().numpy(), ref[3]), 'clusters image %d' % b


@pytest.mark.parametrize('name,shape,grid,images', [
    ('cfg2', (48, 256, 448, 448), (8, 8), (11,)),
    ('cfg3', (16, 256, 224, 224), (8, 8), (0, 15)),
    ('cfg4', (4, 256, 768, 768), (16, 16), (2,)),
    ('cfg5', (24, 384, 224, 224), (8, 16), (5, 23)),
])
def test_full_size_configs_filters_verified_and_spot_parity(dev, oracle, name, shape, grid, images):
  """The per-GPU shapes of BASELINE.json configs 2-5 at FULL size, inputs from the portable
  generator (seed 0x48534700 + cfg), 10 iterations, run under the library's verify switch:
  every filtered E-step of the call (fp16 level, hi-plane / two-half variants, bf16x3,
  exact chains) is re-done by the exact fp32 E-step on the device and must agree on EVERY
  row; plus size-independent properties of the outputs and bit-exact parity of whole images
  against the oracle."""
  import torch
  from hsg_amd import _lib
  from hsg_amd.utils.segsort import common as sc
  B, C, H, W = shape
  K, iters = grid[0] * grid[1], 10
  x = synth.device_embeddings_nchw(synth.SEED_BASE + int(name[3:]), shape, 'iid', dev)
  _lib.verify_collect()
  _lib.verify_enable(True)
  try:
    emb, eloc, labels, cluster, batch = sc.segment_by_kmeans(x, None, list(grid), iterations=iters)
    compared, differing = _lib.verify_collect()
  finally:
    _lib.verify_enable(False)
  n = B * H * W
  assert compared == n * iters, 'the filtered E-step route was not taken (%d rows compared)' % compared
  assert differing == 0, '%d filter-decided labels differ from the exact E-step' % differing
  assert emb.shape == (n, C) and eloc.shape == (n, C + 2)
  assert (emb.norm(dim=1) - 1).abs().max().item() < 1e-5
  assert (eloc.norm(dim=1) - 1).abs().max().item() < 1e-5
  assert torch.equal(batch, torch.arange(B, device=dev).repeat_interleave(H * W))
  # dense ids sorted by (image, cluster): image b+1 starts right after the last id of image b
  per = cluster.view(B, H * W)
  lo, hi = per.min(dim=1).values, per.max(dim=1).values
  assert int(lo[0]) == 0 and torch.equal(lo[1:], hi[:-1] + 1) and int((hi - lo).max()) < K
  assert torch.equal(labels, torch.zeros_like(labels))
  loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
  for b in images:
    ref = oracle.segment_by_kmeans(x[b:b + 1].cpu().numpy(), None, grid, loc, None, iters)
    sl = slice(b * H * W, (b + 1) * H * W)
    assert np.array_equal(emb[sl].cpu().numpy(), ref[0]), 'emb image %d' % b
    assert np.array_equal(eloc[sl].cpu().numpy(), ref[1]), 'emb_loc image %d' % b
    # dense ids: image b owns the ids b*K .. b*K + K-1 when no cluster is empty
    ids = cluster[sl].cpu().numpy()
    assert np.array_equal(ids - ids.min(), ref[3] - ref[3].min()), 'clusters image %d' % b


@pytest.mark.parametrize('filter_kernel', ['', 'regs', 'one', 'two'])
def test_cfg4_end_to_end_c256_grid16_labels_ignore_vs_oracle(dev, oracle, monkeypatch, filter_kernel):
  """BASELINE.json configs[3] route end to end on a reduced map: C = 256 with a 16x16 seed
  grid (K = 256 -> the one-pass fp16 filter in pairs of waves by default; HSGK_WIDE2 = regs / one / two: the
  table-in-registers, four-wave and two-half variants kept for the A/B of DESIGN.md 5a; the cluster-split
  exact-sum M-step), over-segmentation labels + ignore band, mixture and i.i.d. inputs --
  all five outputs bit-exact vs the oracle, every filtered label verified on the device."""
  from hsg_amd import _lib
  from hsg_amd.utils.segsort import common as sc
  if filter_kernel:
    monkeypatch.setenv('HSGK_WIDE2', filter_kernel)
  for flavour, shape, iters in (('iid', (2, 256, 96, 96), 10), ('mixture', (1, 256, 128, 80), 6)):
    B, C, H, W = shape
    x = synth.embeddings_nchw(synth.SEED_BASE + 4, shape, flavour)
    lab = synth.overseg_labels(synth.SEED_BASE + 44, B, H, W, regions=11, ignore_rows=3)
    loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
    _lib.verify_collect()
    _lib.verify_enable(True)
    try:
      got = _run_segkm(dev, x, lab, (16, 16), 255, iters)
      compared, differing = _lib.verify_collect()
    finally:
      _lib.verify_enable(False)
    assert compared == got[0].shape[0] * iters and differing == 0, (compared, differing)
    ref = oracle.segment_by_kmeans(x, lab, (16, 16), loc, 255, iters)
    for nm, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
      assert a.shape == b.shape, nm
      assert np.array_equal(a, b), '%s (%s): %d mismatching elements' % (nm, flavour, int((a != b).sum()))


def test_whole_train_step_vs_reference(dev):
  """one WHOLE training step around a stub backbone -- train.py:165-269 restated in
  tests/util.run_train_step -- through the hsg_amd model-level mirrors (MultiviewResnetFcn's
  clustering half, the prototype exchange x3, the cluster mappings, Hsg.forward with the three
  contrastive losses in ONE E.P^T pass + DMon + centroid contrast, backward to the embeddings)
  against the same step run by the reference's own modules (tests/golden/f14_train_step_full.npz):
  every integer output identical, losses within 1e-4, gradients within 1e-5 of their scale."""
  import torch
  from hsg_amd.models import utils as mu
  from hsg_amd.models.embeddings import resnet_fcn_hsg as em
  from hsg_amd.models.predictions import hsg as pm
  from hsg_amd.utils.segsort import common as sc
  g = util.load('f14_train_step_full')
  inp = util.train_step_inputs(int(g['seed']))
  loc_fn = lambda hw, d: sc.generate_location_features(hw, d, 'float') - 0.5
  out = util.run_train_step(dict(embedding_cls=em.MultiviewClusteringMixin, prediction_cls=pm.Hsg,
                                 model_utils=mu, loc_fn=loc_fn), inp, dev)
  for k in ('image_index', 'cluster_index', 'finehrchy_cluster_index', 'coarsehrchy_cluster_index',
            'finehrchy_mapping_index', 'coarsehrchy_mapping_index', 'n_prototypes'):
    assert np.array_equal(out[k].cpu().numpy(), g[k]), k
  for k in ('img_sim_loss', 'hrchy_group_loss', 'clustering_loss'):
    assert abs(float(out[k].detach()) - float(g[k])) <= 1e-4, (k, float(out[k].detach()), float(g[k]))
  assert abs(float(out['accuracy'].detach()) - float(g['accuracy'])) <= 1e-6
  for k in ('grad', 'g_fine_logits', 'g_coarse_logits', 'g_cent_f'):
    ref = g[k]
    scale = max(float(np.abs(ref).max()), 1e-6)
    err = np.abs(out[k].cpu().numpy() - ref)
    # 'segsort+' forms `same-label sum - own similarity` in fp32 (loss.py:63-66); where the own
    # prototype dominates that sum the difference is summation-order noise in the reference itself,
    # and 1/num carries it into those pixels' gradients.  The reference re-run with its row sums in
    # float64 moves by p50 4.4e-8, p90 7.1e-7, p99 7.2e-6, max 4.8e-5 on this very step (scale 3.2e-2,
    # tests/checkers/train_step_fp32_noise.py) -- the bounds below are that distribution with margin
    q = [float(np.quantile(err, v)) for v in (0.5, 0.9, 0.99)]
    print(k, 'scale %.3e p50 %.3e p90 %.3e p99 %.3e max %.3e' % (scale, q[0], q[1], q[2], err.max()))
    assert q[0] <= 1e-5 * scale + 1e-8 and q[1] <= 1e-4 * scale + 1e-8, (k, q, scale)
    assert q[2] <= 1e-3 * scale + 1e-8 and err.max() <= 1e-2 * scale + 1e-8, (k, q, float(err.max()), scale)


def test_patch_reference_rebinds_a_reference_shaped_package(dev, tmp_path, monkeypatch):
  """hsg_amd.patch_reference() on a package laid out like twke18/HSG (the real one is not on the GPU
  box): module functions, model methods and Hsg.losses are rebound, and a patched call runs on libhsgk."""
  import importlib
  import sys
  import torch
  import hsg_amd
  root = tmp_path / 'fakehsg'
  for sub in ('', 'utils', 'utils/segsort', 'utils/general', 'models', 'models/embeddings', 'models/predictions'):
    (root / sub).mkdir(parents=True, exist_ok=True)
    (root / sub / '__init__.py').write_text('')
  (root / 'utils/segsort/common.py').write_text('def segment_by_kmeans(*a, **k):\n  raise RuntimeError("reference path")\n'
                                                'def calculate_prototypes_from_labels(*a, **k):\n  raise RuntimeError("reference path")\n')
  (root / 'utils/general/common.py').write_text('def normalize_embedding(*a, **k):\n  raise RuntimeError("reference path")\n')
  (root / 'models/embeddings/resnet_fcn_hsg.py').write_text(
      'import fakehsg.utils.segsort.common as segsort_common\n'
      'class ResnetFcn:\n  def generate_clusters(self):\n    return "reference"\n'
      'class MultiviewResnetFcn(ResnetFcn):\n  pass\n')
  (root / 'models/predictions/hsg.py').write_text('class Hsg:\n  def losses(self, datas, targets={}):\n    return "reference"\n')
  monkeypatch.syspath_prepend(str(tmp_path))
  done = hsg_amd.patch_reference('fakehsg')
  assert 'fakehsg.utils.segsort.common.segment_by_kmeans' in done
  assert 'fakehsg.models.embeddings.resnet_fcn_hsg.MultiviewResnetFcn.generate_clusters' in done
  assert 'fakehsg.models.predictions.hsg.Hsg.losses' in done
  model = importlib.import_module('fakehsg.models.embeddings.resnet_fcn_hsg')
  x = torch.from_numpy(synth.embeddings_nchw(5, (1, 8, 12, 12), 'iid')).to(dev)
  out = model.segsort_common.segment_by_kmeans(x, None, [2, 2], iterations=2)       # the alias sees the rebinding
  assert out[0].shape == (144, 8) and out[3].max().item() == 3
  for name in list(sys.modules):
    if name.startswith('fakehsg'):
      del sys.modules[name]


def test_train_step_slice_vs_reference(dev):
  """k-means -> batch prototype table -> two SegSort losses ->
  backward to the NCHW embeddings, against the reference running the same
  calls on CPU (tests/golden/f9_train_step.npz)."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort.loss import SegSortLoss
  from hsg_amd.models import utils as mu
  g = util.load('f9_train_step')
  shape = tuple(int(v) for v in g['shape'])
  grid = [int(v) for v in g['grid']]
  seed = int(g['seed'])
  x = torch.from_numpy(synth.embeddings_nchw(seed, shape, 'mixture')).to(dev).requires_grad_(True)
  lab = torch.from_numpy(synth.overseg_labels(int(g['label_seed']), shape[0], shape[2], shape[3],
                                              regions=6, ignore_rows=2, ignore_index=255)).to(dev)
  emb, emb_loc, labels, cidx, bidx = sc.segment_by_kmeans(x, lab, grid, ignore_index=255, iterations=6)
  zeros = torch.zeros_like(labels)
  protos, protos_loc, psem, pinst, pbatch, upd = mu.gather_clustering_and_update_prototypes(
      [emb], [emb_loc], [cidx], [bidx], [labels], [zeros], dev)
  assert protos[0].shape[0] == int(g['n_protos'])
  assert np.array_equal(upd[0].cpu().numpy(), g['upd'].astype(np.int64))
  loss_a = SegSortLoss(16, 'segsort+')(emb, labels, upd[0], protos[0], psem[0])
  loss_b = SegSortLoss(10, 'segsort')(emb_loc, labels, upd[0], protos_loc[0], psem[0])
  assert abs(loss_a.item() - float(g['loss_a'])) <= 1e-4
  assert abs(loss_b.item() - float(g['loss_b'])) <= 1e-4
  (loss_a + 0.5 * loss_b).backward()
  got = x.grad.cpu().numpy().reshape(-1)[::11]
  assert np.abs(got - g['grad']).max() <= 2e-5 * max(float(g['grad_absmax']), 1.0)


@pytest.mark.parametrize('n,c,P,k', [(500, 48, 97, 3), (3000, 128, 700, 5), (2100, 256, 64, 20),
                                     (70, 32, 40, 32)])
def test_top_k_ranking_vs_torch(dev, oracle, n, c, P, k):
  """eval.py:9-52: same accuracy and retrieved labels as the reference's
  mm + argsort formulation (fp32 ATen on the GPU) on tie-free data."""
  import torch
  from hsg_amd.utils.segsort import eval as ev
  e = torch.from_numpy(oracle.normalize_embedding(synth.gaussish(3 + n, n * c).reshape(n, c))).to(dev)
  p = torch.from_numpy(oracle.normalize_embedding(synth.gaussish(5 + P, P * c).reshape(P, c))).to(dev)
  plab = torch.from_numpy((synth.hash_u64(7 + P, P) % np.uint64(9)).astype(np.int64)).to(dev)
  lab = torch.from_numpy((synth.hash_u64(9 + n, n) % np.uint64(9)).astype(np.int64)).to(dev)
  acc, labs = ev.top_k_ranking(e, lab, p, plab, k)
  aff = torch.mm(e, p.t())
  idx = torch.argsort(aff, 1, descending=True)[:, :k]
  ref_labs = plab[idx.reshape(-1)].view(-1, k)
  ref_acc = torch.eq(lab.view(-1, 1), ref_labs).float().mean()
  got_idx, got_val = ev.top_k_indices(e, p, k)
  # scores agree to rounding and are sorted; indices agree wherever the gap to
  # the next score exceeds the fp32 dot-product noise
  ref_val = torch.gather(aff, 1, idx)
  assert (got_val - ref_val).abs().max().item() <= 2e-6
  gap_ok = torch.ones_like(idx, dtype=torch.bool)
  srt = torch.sort(aff, 1, descending=True).values[:, :k + 1]
  gap_ok &= (srt[:, :k] - srt[:, 1:k + 1]) > 1e-5 if srt.shape[1] > k else gap_ok
  if k > 1:
    gap_ok[:, 1:] &= (srt[:, :k - 1] - srt[:, 1:k]) > 1e-5
  assert torch.equal(got_idx[gap_ok], idx[gap_ok])
  assert abs(acc.item() - ref_acc.item()) <= 2e-3
  assert (labs == ref_labs)[gap_ok].all()
  maj = ev.majority_label_from_topk(labs, 9)
  assert maj.shape == (n,)


def test_transformer_clustering_tail_vs_reference_golden(dev, oracle):
  """a11: hsgk_cluster_topk through the Python mirror against the reference's own
  TransformerClustering.forward tail (tests/golden/f10_cluster_tail.npz) -- selection
  identical, logits bit-exact vs the oracle and <= 1e-5 vs the reference, gradients of
  all three inputs vs the reference's autograd."""
  import torch
  from hsg_amd.models.embeddings import hierarchy as hz
  g = util.load('f10_cluster_tail')
  seed = int(g['seed'])
  B, C, tl, sl, k = (int(v) for v in g['shape'])
  cen_np = synth.gaussish(seed, B * C * tl).reshape(B, C, tl).copy()
  nod_np = synth.gaussish(seed + 1, B * C * sl).reshape(B, C, sl).copy()
  cen = torch.from_numpy(cen_np).to(dev).requires_grad_(True)
  nod = torch.from_numpy(nod_np).to(dev).requires_grad_(True)
  cfe = cen * 0.5 + 1.0
  c_sel, cf_sel, logits, order = hz.transformer_clustering_tail(cen, cfe, nod, k)
  assert np.array_equal(c_sel.detach().cpu().numpy(), g['c_sel'])
  assert np.array_equal(cf_sel.detach().cpu().numpy(), g['cf_sel'])
  assert np.abs(logits.detach().cpu().numpy() - g['logits']).max() <= 1e-5
  o_c, o_cf, o_l, o_ord = oracle.transformer_clustering_tail(cen_np, cen_np * np.float32(0.5) + np.float32(1.0),
                                                             nod_np, k)
  assert np.array_equal(order.cpu().numpy(), o_ord)
  assert np.array_equal(logits.detach().cpu().numpy(), o_l)          # same C1 chain: bit-exact
  T = lambda s_, shape: torch.from_numpy(synth.gaussish(s_, int(np.prod(shape))).reshape(shape).copy()).to(dev)
  ((c_sel * T(seed + 2, (B, C, k))).sum() + (cf_sel * T(seed + 3, (B, C, k))).sum()
   + (logits * T(seed + 4, (B, k, sl))).sum()).backward()
  assert np.abs(cen.grad.cpu().numpy() - g['g_cen']).max() <= 1e-4
  assert np.abs(nod.grad.cpu().numpy() - g['g_nod']).max() <= 1e-4


def test_transformer_clustering_tail_ties_and_full_permutation(dev, oracle):
  """k == tl (the reference's configuration: the top-k is a permutation sorted by maximum
  activation), duplicated queries (exact ties -> lower index first) at a larger shape."""
  import torch
  from hsg_amd.models.embeddings import hierarchy as hz
  B, C, tl, sl = 4, 256, 64, 256
  cen = synth.gaussish(991, B * C * tl).reshape(B, C, tl).copy()
  cen[:, :, 7] = cen[:, :, 3]                                         # exact tie
  cen[:, :, 40] = cen[:, :, 3]
  nod = synth.gaussish(992, B * C * sl).reshape(B, C, sl).copy()
  cfe = synth.gaussish(993, B * C * tl).reshape(B, C, tl).copy()
  got = hz.transformer_clustering_tail(torch.from_numpy(cen).to(dev), torch.from_numpy(cfe).to(dev),
                                       torch.from_numpy(nod).to(dev), tl)
  ref = oracle.transformer_clustering_tail(cen, cfe, nod, tl)
  for a, b in zip(got, ref):
    assert np.array_equal(a.cpu().numpy(), b)
  o = got[3].cpu().numpy()
  for b in range(B):
    assert sorted(o[b].tolist()) == list(range(tl))
    p3, p7, p40 = (int(np.where(o[b] == q)[0][0]) for q in (3, 7, 40))
    assert p3 < p7 < p40 and p7 == p3 + 1 and p40 == p7 + 1


def test_set_segsort_loss_vs_golden_and_oracle(dev, oracle):
  """n3: SetSegSortLoss through the loss kernels' set mode (class bit masks) against the
  reference's own module (tests/golden/f11_set_segsort_loss.npz): loss within 1e-4,
  per-pixel nll, both gradients; and against the oracle's affinity formulation."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort.loss import SetSegSortLoss
  g = util.load('f11_set_segsort_loss')
  n, c, P, nc = (int(v) for v in g['shape'])
  e_raw, inst_np, sem_np, psem_np = util.set_loss_inputs(int(g['seed']), n, c, P, nc)
  e_np = oracle.normalize_embedding(e_raw)
  inst, sem, psem = (torch.from_numpy(a).to(dev) for a in (inst_np, sem_np, psem_np))
  for kappa in (10, 16):
    for mode, tag in (('segsort+', 'plus'), ('segsort', 'plain')):
      key = 'k%d_%s' % (kappa, tag)
      e = torch.from_numpy(e_np).to(dev).requires_grad_(True)
      proto = sc.calculate_prototypes_from_labels(e, inst, P)
      pp = proto.detach().clone().requires_grad_(True)
      loss = SetSegSortLoss(kappa, mode)(e, sem, inst, pp, psem)
      loss.backward()
      assert abs(loss.item() - float(g[key + '_loss'])) <= 1e-4
      nll = SetSegSortLoss(kappa, mode, reduction='none')(e.detach(), sem, inst, proto.detach(), psem)
      assert nll.shape == (n, 1)
      assert np.abs(nll.view(-1).cpu().numpy() - g[key + '_nll']).max() <= 1e-3   # see the CPU test
      assert np.abs(e.grad.cpu().numpy()[::7] - g[key + '_gemb']).max() <= 2e-6
      assert np.abs(pp.grad.cpu().numpy() - g[key + '_gproto']).max() <= 2e-5
      ref = oracle.set_segsort_nll(e_np, sem_np, inst_np, proto.detach().cpu().numpy(), psem_np,
                                   float(kappa), mode)
      assert np.abs(nll.view(-1).cpu().numpy() - ref).max() <= 1e-3
      assert abs(loss.item() - ref.mean()) <= 1e-4
  with pytest.raises(ValueError):
    SetSegSortLoss()(e.detach(), -sem, inst, proto.detach(), psem)


def test_segsort_model_predictions_losses_multiset_vs_reference(dev):
  """Rows n1 / n3 at the model level, against the reference's own `Segsort` module and glue
  (tests/golden/f15_segsort_model.npz): nearest-neighbour prediction against a memory bank (top-20
  retrieval + majority vote: every label identical), the three SegSort losses with backward,
  `gather_multiset_labels_per_batch_by_nearest_neighbor` (identical multi-hot labels), and
  SetSegSortLoss with 90 classes (two 63-bit mask words)."""
  import torch
  from hsg_amd.models import utils as mu
  from hsg_amd.models.predictions import segsort as sm
  from hsg_amd.utils.segsort.loss import SetSegSortLoss
  g = util.load('f15_segsort_model')
  inp = util.segsort_inputs(int(g['seed']))
  T = lambda k: torch.from_numpy(inp[k]).to(dev)
  model = sm.Segsort(util.segsort_config())
  datas = {'cluster_embedding': T('emb').requires_grad_(True), 'cluster_embedding_with_loc': T('emb_loc'),
           'cluster_index': T('cidx'), 'cluster_semantic_label': T('sem'), 'cluster_instance_label': T('inst'),
           'cluster_batch_index': T('bidx')}
  targets = {'semantic_memory_prototype': T('mem'), 'semantic_memory_prototype_label': T('mem_lab'),
             'prototype': T('protos').requires_grad_(True), 'prototype_semantic_label': T('psem'),
             'prototype_batch_index': T('pbatch'), 'semantic_tag': T('tags'), 'prototype_semantic_tag': T('ptags')}
  out = model(datas, targets, with_loss=True, with_prediction=True)
  assert np.array_equal(out['semantic_prediction'].cpu().numpy(), g['pred'])
  assert np.array_equal(out['semantic_score'].cpu().numpy(), g['topk'])
  for k, ref in (('sem_ann_loss', 'sem_ann'), ('sem_occ_loss', 'sem_occ'), ('img_sim_loss', 'img_sim')):
    assert abs(float(out[k].detach()) - float(g[ref])) <= 1e-4, (k, float(out[k].detach()), float(g[ref]))
  assert abs(float(out['accuracy'].detach()) - float(g['acc'])) <= 1e-6
  (out['sem_ann_loss'] + out['sem_occ_loss'] + out['img_sim_loss']).backward()
  for got, ref in ((datas['cluster_embedding'].grad, g['g_emb']), (targets['prototype'].grad, g['g_protos'])):
    scale = float(np.abs(ref).max())
    err = np.abs(got.cpu().numpy() - ref)
    assert np.quantile(err, 0.99) <= 1e-4 * scale and err.max() <= 1e-2 * scale, (float(err.max()), scale)
  multi = mu.gather_multiset_labels_per_batch_by_nearest_neighbor(
      T('emb'), T('protos'), T('psem'), T('bidx'), T('pbatch'), num_classes=int(inp['num_classes']), top_k=3,
      threshold=0.3)
  assert np.array_equal(multi.cpu().numpy(), g['multi'])
  # SetSegSortLoss, 90 classes
  e, inst, sem, psem = util.set_loss_inputs(int(g['seed']) + 7, 400, 24, 37, 90)
  et = torch.from_numpy(e).to(dev)
  et = (et / et.norm(dim=1, keepdim=True)).requires_grad_(True)
  for mode, tag in (('segsort+', 'plus'), ('segsort', 'plain')):
    pt = torch.from_numpy(g['set90_proto']).to(dev).requires_grad_(True)
    loss = SetSegSortLoss(12, mode)(et, torch.from_numpy(sem).to(dev), torch.from_numpy(inst).to(dev), pt,
                                    torch.from_numpy(psem).to(dev))
    ge, gp = torch.autograd.grad(loss, [et, pt])
    assert abs(loss.item() - float(g['set90_%s_loss' % tag])) <= 1e-4
    assert np.abs(ge.cpu().numpy() - g['set90_%s_gemb' % tag]).max() <= 2e-5 * max(1.0, float(np.abs(g['set90_%s_gemb' % tag]).max()) * 100)
    assert np.abs(gp.cpu().numpy() - g['set90_%s_gproto' % tag]).max() <= 2e-4


def test_inference_pieces_vs_reference_golden(dev, oracle):
  """n2: find_majority_label_index (histogram / argmax / select kernels) and the
  overlap-averaged patch accumulation (normalise + accumulate + divide kernels) against
  the reference (tests/golden/f12_inference.npz) and bit-exact against the oracle."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort import inference as inf
  g = util.load('f12_inference')
  sem, clu, crops, corners, (C, H, W), _, _ = util.inference_inputs(g)
  sel, maj = sc.find_majority_label_index(torch.from_numpy(sem).to(dev), torch.from_numpy(clu).to(dev))
  assert sel.dtype == torch.int64 and sel.dim() == 2 and sel.shape[1] == 1
  assert np.array_equal(maj.cpu().numpy(), g['maj'])
  assert np.array_equal(sel.cpu().numpy(), g['sel'].astype(np.int64))
  crop_h, crop_w, stride_h, stride_w = (int(v) for v in g['ov_shape'][3:7])
  assert np.array_equal(inf.patch_end_indices(H, crop_h, stride_h), g['patch_ind_h'])
  assert np.array_equal(inf.patch_end_indices(W, crop_w, stride_w), g['patch_ind_w'])
  avg = inf.OverlapAverager(C, H, W, dev)
  for crop, (sh, sw) in zip(crops, corners):
    avg.add(torch.from_numpy(crop).to(dev).unsqueeze(0), sh, sw)
  from hsg_amd import _lib
  with pytest.raises(_lib.HsgkError):
    avg.add(torch.from_numpy(crops[0]).to(dev), H, 0)                              # outside the canvas
  canvas = avg.result()
  assert canvas.shape == (1, C, H, W)
  got = canvas[0].cpu().numpy()
  assert np.abs(got - g['canvas']).max() <= FTOL
  assert np.array_equal(got, oracle.overlap_average(crops, corners, C, H, W))     # same C1 chain and order


def test_inference_pipeline_end_to_end_vs_oracle(dev, oracle, tmp_path):
  """prototype.py:141-208 as a whole on a small image: overlap-averaged embeddings ->
  full-resolution segment_by_kmeans (ignore band from the padding) -> prototypes ->
  majority labels -> .npy memory bank -> reload; every stage against the oracle."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort import inference as inf
  from hsg_amd.utils.segsort import others
  g = util.load('f12_inference')
  _, _, crops, corners, (C, H, W), _, _ = util.inference_inputs(g)
  avg = inf.OverlapAverager(C, H, W, dev)
  for crop, (sh, sw) in zip(crops, corners):
    avg.add(torch.from_numpy(crop).to(dev), sh, sw)
  emb_full = avg.result()
  fake = np.zeros((1, H, W), np.int64)
  fake[:, H - 6:, :] = 255                                   # padded rows are ignored by the clustering
  true_sem = (synth.hash_u64(777, H * W) % np.uint64(4)).astype(np.int64).reshape(1, H, W)
  loc = sc.generate_location_features((H, W), dev, 'float') - 0.5     # torch's linspace bits are input data
  out = sc.segment_by_kmeans(emb_full, torch.from_numpy(fake).to(dev), [3, 4], local_features=loc,
                             ignore_index=255, iterations=5)
  ref = oracle.segment_by_kmeans(emb_full.cpu().numpy(), fake, (3, 4), local_features=loc.cpu().numpy(),
                                 ignore_index=255, iterations=5)
  for a, b in zip(out, ref):
    assert np.array_equal(a.cpu().numpy(), b)
  emb, _, _, cidx, _ = out
  protos = sc.calculate_prototypes_from_labels(emb, cidx)
  assert np.array_equal(protos.cpu().numpy(), oracle.calculate_prototypes_from_labels(ref[0], ref[3]))
  keep = torch.from_numpy(fake.reshape(-1) != 255).to(dev)
  sem_kept = torch.from_numpy(true_sem.reshape(-1)).to(dev)[keep]
  sel, maj = sc.find_majority_label_index(sem_kept, cidx)
  o_sel, o_maj = oracle.find_majority_label_index(sem_kept.cpu().numpy(), ref[3])
  assert np.array_equal(maj.cpu().numpy(), o_maj) and np.array_equal(sel.cpu().numpy(), o_sel)
  others.save_prototypes(str(tmp_path / 'img0.npy'), protos, maj)
  bank_p, bank_l = others.load_memory_banks(str(tmp_path))
  assert np.array_equal(bank_p.numpy(), protos.cpu().numpy()) and np.array_equal(bank_l.numpy(), o_maj)


def test_dmon_affinity_graph_and_loss_vs_reference_golden(dev, oracle):
  """n4: hsgk_knn_affinity through the graph mirror against the reference's
  affinity_matrix_as_attention (tests/golden/f13_dmon_graph.npz) and the oracle; DMonLoss
  value and logit gradients against the reference's autograd."""
  import torch
  from hsg_amd.utils.graph import common as gc
  from hsg_amd.utils.graph import loss as gl
  g = util.load('f13_dmon_graph')
  B, C, N, K, knn = (int(v) for v in g['shape'])
  x, pad, seg, logits = util.graph_inputs(int(g['seed']), B, C, N, K)
  xt, padt, segt = (torch.from_numpy(a).to(dev) for a in (x, pad, seg))
  a_knn = gc.affinity_matrix_as_attention(xt, padt, segt, knn, True, True, concentration=5)
  assert a_knn.shape == (B, N, N) and a_knn.dtype == torch.float32
  assert np.array_equal(a_knn.cpu().numpy().astype(np.uint8), g['adj_knn'])
  a_all = gc.affinity_matrix_as_attention(xt, padt)
  assert np.array_equal(a_all.cpu().numpy().astype(np.uint8), g['adj_all'])
  a_val = gc.affinity_matrix_as_attention(xt, padt, segt, 3, False, False)
  ref_val = oracle.affinity_matrix_as_attention(x, pad, seg, 3, remove_self_loop=False, binarize=False)
  assert np.array_equal(a_val.cpu().numpy() > 0, g['adj_val'] > 0)
  assert np.abs(a_val.cpu().numpy() - ref_val).max() <= 1e-5 * ref_val.max()      # expf ulps only
  # a caller-evaluated kernel function: same graph through the affinity_in path
  a_fn = gc.affinity_matrix_as_attention(xt, padt, segt, knn, True, True,
                                         kernel_fn=lambda t: gc.exp_inner_product_kernel(t, 5))
  assert np.array_equal(a_fn.cpu().numpy().astype(np.uint8), g['adj_knn'])
  keep = [0, 1, 3, 4]
  lg = torch.from_numpy(logits).to(dev).requires_grad_(True)
  d, c = gl.DMonLoss(adj_knn=knn)(torch.softmax(lg[keep], 1), xt[keep], padt[keep], segt[keep])
  (d + 0.5 * c).backward()
  assert abs(d.item() - float(g['dmon'])) <= 1e-5 and abs(c.item() - float(g['collapse'])) <= 1e-5
  assert np.abs(lg.grad.cpu().numpy() - g['g_logits']).max() <= 1e-6


def test_hierarchical_dmon_loss_vs_reference_golden(dev):
  """HierarchicalDMonLoss (reference graph/loss.py:148-231) with two levels -- the first on the binary k-NN graph
  (fused pooling kernels), the second on the adjacency pooled with the first level's masked probabilities (it
  carries a gradient: the formula path) -- against the reference's values and logit gradients
  (tests/golden/f17_hier_dmon.npz, tools/gen_golden.py f17)."""
  import torch
  from hsg_amd.utils.graph import loss as gl
  g = util.load('f17_hier_dmon')
  B, C, N, K1, K2, knn = (int(v) for v in g['shape'])
  x, pad, seg, logits1 = util.graph_inputs(int(g['seed']), B + 1, C, N, K1)
  keep = [0, 1, 3, 4]
  T = lambda a: torch.from_numpy(a).to(dev)
  logits2 = synth.gaussish(int(g['seed']) + 5, B * K2 * K1).reshape(B, K2, K1).copy()
  l1, l2 = T(logits1[keep]).requires_grad_(True), T(logits2).requires_grad_(True)
  dm, co = gl.HierarchicalDMonLoss(adj_knn=knn)([torch.softmax(l1, 1), torch.softmax(l2, 1)], T(x[keep]),
                                                [T(pad[keep]), T(g['pad2'])], T(seg[keep]))
  (dm[0] + 0.5 * co[0] + 2.0 * dm[1] + 0.25 * co[1]).backward()
  for i in range(2):
    assert abs(dm[i].item() - float(g['dmon'][i])) <= 1e-5 and abs(co[i].item() - float(g['collapse'][i])) <= 1e-5
  assert np.abs(l1.grad.cpu().numpy() - g['g_logits1']).max() <= 2e-6
  assert np.abs(l2.grad.cpu().numpy() - g['g_logits2']).max() <= 2e-6


def test_cityscapes_twins_of_the_model_classes(dev):
  """`hsg_cs.Hsg` scores the grouping logits against a k-NN graph over all nodes of an image row (no per-view
  segments, hsg_cs.py:173-175): HsgCs's DMon terms == Hsg's with a constant segment label, != with real views;
  the embedding mix-ins of resnet_fcn_hsg_cs.py only switch the pad length to the call's maximum."""
  import types
  import torch
  from hsg_amd.models.predictions import hsg as pm
  from hsg_amd.models.embeddings import resnet_fcn_hsg as em
  from hsg_amd.utils.graph import loss as gl
  B, C, N = 3, 24, 40
  g = torch.Generator(device=dev).manual_seed(7)
  nodes = torch.nn.functional.normalize(torch.randn((B, C, N), device=dev, generator=g), dim=1)
  datas = {'nd_prototype': nodes, 'nd_prototype_padding_mask': torch.zeros((B, N), dtype=torch.bool, device=dev),
           'nd_prototype_batch_index': (torch.arange(N, device=dev) % 2).expand(B, N).contiguous(),
           'finehrchy_nd_prototype_grouping_logit': torch.randn((B, 6, N), device=dev, generator=g),
           'coarsehrchy_nd_prototype_grouping_logit': torch.randn((B, 3, N), device=dev, generator=g)}
  loss = gl.DMonLoss(adj_knn=4)
  per_view = types.SimpleNamespace(dmon_loss=loss, dmon_graph_per_view=True)
  whole = types.SimpleNamespace(dmon_loss=loss, dmon_graph_per_view=pm.HsgCs.dmon_graph_per_view)
  a = pm._dmon_terms(per_view, datas)
  b2 = pm._dmon_terms(whole, datas)
  one = dict(datas, nd_prototype_batch_index=torch.zeros_like(datas['nd_prototype_batch_index']))
  c = pm._dmon_terms(per_view, one)
  assert abs(b2.item() - c.item()) <= 1e-6 and abs(a.item() - b2.item()) > 1e-4
  assert pm.hsg_cs.__name__ == 'hsg_cs' and issubclass(pm.HsgCs, pm.Hsg)
  assert em.ClusteringMixinCs.dynamic_max_num_clusters and em.MultiviewClusteringMixinCs.dynamic_max_num_clusters
  assert em.MultiviewClusteringMixinCs.generate_clusters is em.MultiviewClusteringMixin.generate_clusters
  # the stage-1 / inference model (resnet_fcn.py:90-148): k-means only, six dict entries
  from hsg_amd.models.embeddings import resnet_fcn as s1
  from hsg_amd.utils.segsort import common as sc
  stub = types.SimpleNamespace(label_divisor=255, semantic_ignore_index=255, kmeans_num_clusters=[3, 3], kmeans_iterations=4)
  x = torch.from_numpy(synth.embeddings_nchw(11, (2, 16, 20, 24), 'mixture')).to(dev)
  over = synth.overseg_labels(12, 2, 20, 24, regions=4, ignore_rows=2, ignore_index=255)
  sem = torch.from_numpy(np.where(over == 255, 255, over % 3).astype(np.int64)).to(dev)
  inst = torch.from_numpy(np.where(over == 255, 0, over // 3).astype(np.int64)).to(dev)
  out = s1.generate_clusters(stub, x, sem, inst)
  lab = sem * 255 + inst
  ign = lab.max() + 1
  ref = sc.segment_by_kmeans(x, lab.masked_fill(sem == 255, ign), [3, 3], ignore_index=ign, iterations=4)
  assert sorted(out) == ['cluster_batch_index', 'cluster_embedding', 'cluster_embedding_with_loc', 'cluster_index',
                         'cluster_instance_label', 'cluster_semantic_label']
  assert torch.equal(out['cluster_embedding'], ref[0]) and torch.equal(out['cluster_index'], ref[3])
  assert torch.equal(out['cluster_semantic_label'] * 255 + out['cluster_instance_label'], ref[2])


def test_ncut_loss_vs_reference_golden(dev):
  """NCutLoss / ncut_pool_loss (reference graph/loss.py:234-345) against the reference's values and logit gradient
  (tests/golden/f18_ncut.npz)."""
  import torch
  from hsg_amd.utils.graph import loss as gl
  g = util.load('f18_ncut')
  B, C, N, K, knn = (int(v) for v in g['shape'])
  x, pad, seg, logits = util.graph_inputs(int(g['seed']), B, C, N, K)
  keep = [0, 1, 3, 4]
  T = lambda a: torch.from_numpy(a).to(dev)
  lg = T(logits[keep]).requires_grad_(True)
  nc, se = gl.NCutLoss(adj_knn=knn)(lg, T(x[keep]), T(pad[keep]), T(seg[keep]))
  (nc + 0.5 * se).backward()
  assert abs(nc.item() - float(g['ncut'])) <= 1e-5 and abs(se.item() - float(g['self_loss'])) <= 1e-5
  assert np.abs(lg.grad.cpu().numpy() - g['g_logits']).max() <= 2e-6


@pytest.mark.parametrize('B,N,K,masked', [(3, 256, 8, True), (2, 8, 4, False), (4, 100, 32, True), (1, 1000, 5, True)])
def test_dmon_pool_fused_kernels_vs_reference_formula(dev, B, N, K, masked):
  """hsgk_dmon_pool_fwd / _bwd (an adjacency without gradient: DMonLoss) against the reference's chain of batched
  GEMMs (loss.py:62-94, the path `dmon_pool_loss` keeps for adjacencies that carry a gradient) evaluated in
  float64: both loss values and the gradient w.r.t. the assignments; masked rows get exactly zero gradient."""
  import torch
  from hsg_amd.utils.graph import loss as gl
  g = torch.Generator(device=dev).manual_seed(B * 1000 + N + K)
  adj = (torch.rand((B, N, N), device=dev, generator=g) < 0.1).float()
  adj[:, torch.arange(N), torch.arange(N)] = 0.0
  adj[:, 0, 1] = 1.0                                                   # (never an empty graph)
  logits = torch.randn((B, N, K), device=dev, generator=g)
  mask = (torch.rand((B, N), device=dev, generator=g) < 0.9) if masked else None
  if mask is not None:
    mask[:, :2] = True
  s1 = logits.clone().requires_grad_(True)
  d1, c1 = gl.dmon_pool_loss(None, adj, s1, mask, softmax=True)
  (d1 + 0.7 * c1).backward()
  s2 = logits.double().requires_grad_(True)
  a2 = adj.double().requires_grad_(True)                               # a gradient-carrying adjacency: the formula path
  d2, c2 = gl.dmon_pool_loss(None, a2, s2, mask, softmax=True)
  (d2 + 0.7 * c2).backward()
  assert abs(d1.item() - d2.item()) <= 1e-5 and abs(c1.item() - c2.item()) <= 1e-5
  scale = max(s2.grad.abs().max().item(), 1e-12)
  assert (s1.grad.double() - s2.grad).abs().max().item() <= 2e-5 * scale
  if mask is not None:
    # (through the softmax the masked rows' logit gradient is zero as well)
    assert float(s1.grad[~mask].abs().max()) == 0.0


def test_dmon_affinity_graph_larger_vs_oracle(dev, oracle):
  """256 nodes, 3 segments, knn 10, ties from duplicated nodes."""
  import torch
  from hsg_amd.utils.graph import common as gc
  B, C, N = 3, 64, 256
  x = synth.gaussish(4242, B * C * N).reshape(B, C, N).copy()
  x /= np.sqrt((x * x).sum(1, keepdims=True))
  x[:, :, 17] = x[:, :, 3]                                   # duplicate node: tied affinities
  pad = np.zeros((B, N), bool)
  pad[1, 200:] = True
  seg = (synth.hash_u64(4243, B * N) % np.uint64(3)).astype(np.int64).reshape(B, N)
  seg[:, 17] = seg[:, 3]
  got = gc.affinity_matrix_as_attention(torch.from_numpy(x.astype(np.float32)).to(dev),
                                        torch.from_numpy(pad).to(dev), torch.from_numpy(seg).to(dev), 10)
  ref = oracle.affinity_matrix_as_attention(x.astype(np.float32), pad, seg, 10)
  assert np.array_equal(got.cpu().numpy(), ref)


@pytest.mark.parametrize('B,HW,C,K', [(2, 5000, 256, 64), (3, 3000, 384, 128), (1, 700, 30, 5), (2, 900, 1030, 9),
                                      (1, 600, 513, 20), (2, 400, 1, 3),
                                      # the matrix-core update (K <= 64, 32-column blocks + tail columns): one and two
                                      # cluster blocks, 4 / 3 / 8 column blocks, tails of 2 / 6 / 39 / 0 columns
                                      (2, 3000, 128, 16), (1, 4100, 100, 40), (3, 2500, 293, 64), (2, 2300, 62, 33)])
@pytest.mark.parametrize('route', ['lds', 'mfma'])
def test_exact_sum_mstep_full_and_incremental(dev, oracle, B, HW, C, K, route, monkeypatch):
  """C2x M-step (hsgk_lloyd_mstep_exact): from scratch == oracle exact sums (centroids
  bit-exact) with adversarial labels (every strip touches every cluster; K * d beyond the
  LDS table at C=384/K=128 -> strip kernel with several slot rounds); then two
  incremental updates (30 % / 1 % of the rows relabelled, some clusters emptied) give the
  SAME int64 sums and centroids as from-scratch passes."""
  import torch
  from hsg_amd import _lib
  if route == 'mfma' and (K > 64 or C + 2 < 32):
    pytest.skip('the matrix-core update covers K <= 64 and rows of at least 32 columns')
  monkeypatch.setenv('HSGK_MSTEP', route)                   # (read per call: sums_fx.hip)
  D = C + 2
  n = B * HW
  x = oracle.normalize_embedding(synth.gaussish(500 + C, n * D).reshape(n, D))
  lab0 = (synth.hash_u64(501 + K, n) % np.uint64(K)).astype(np.int32)
  lab1 = lab0.copy()
  m = (synth.hash_u64(502, n) % np.uint64(10)) < 3
  lab1[m] = (synth.hash_u64(503, n)[m] % np.uint64(K)).astype(np.int32)
  lab1[lab1 == K - 1] = 0                                          # empty the last cluster
  lab2 = lab1.copy()
  m2 = (synth.hash_u64(504, n) % np.uint64(100)) == 0
  lab2[m2] = K - 1
  L = _lib.lib()
  xt = torch.from_numpy(x).to(dev)
  wsb = L.hsgk_lloyd_workspace_bytes(B, HW, D, K)
  ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)

  def mstep(prev, cur, sums):
    cent = torch.empty((B, K, D), dtype=torch.float32, device=dev)
    _lib.check(L.hsgk_lloyd_mstep_exact(xt.data_ptr(), B, HW, D, K, prev.data_ptr() if prev is not None else None,
                                        cur.data_ptr(), sums.data_ptr(), cent.data_ptr(), ws.data_ptr(), wsb,
                                        _lib.stream_ptr()))
    return cent
  t = lambda a: torch.from_numpy(a).to(dev)
  sums = torch.empty((B, K, D), dtype=torch.int64, device=dev)
  l0, l1, l2 = t(lab0), t(lab1), t(lab2)
  for step, (prev, cur, lab_np) in enumerate(((None, l0, lab0), (l0, l1, lab1), (l1, l2, lab2))):
    cent = mstep(prev, cur, sums)
    fresh = torch.empty_like(sums)
    cent_fresh = mstep(None, cur, fresh)
    assert torch.equal(sums, fresh), 'incremental sums differ from a from-scratch pass (step %d)' % step
    assert torch.equal(cent, cent_fresh)
    for b in range(B):
      ref = oracle.calculate_prototypes_from_labels(x[b * HW:(b + 1) * HW], lab_np[b * HW:(b + 1) * HW].astype(np.int64),
                                                    K, exact_sums=True)
      assert np.array_equal(cent[b].cpu().numpy(), ref), 'centroids vs oracle (step %d, image %d)' % (step, b)


@pytest.mark.gpu
@pytest.mark.parametrize('seed,B,n,K,C,multiview,M', [(1, 3, 900, 7, 24, False, 32), (2, 4, 2500, 16, 64, True, 64),
                                                      (3, 2, 300, 5, 130, True, None), (4, 5, 1200, 9, 16, False, None),
                                                      (5, 1, 50, 3, 8, False, 8)])
def test_padded_prototype_tables_one_pass_vs_aten(dev, seed, B, n, K, C, multiview, M):
  """calculate_kmeans_prototypes on hsgk_pad_prototype_tables (one pass: scans, placement, masks, labels, per-pixel
  rank / image) against the ATen formulation it replaced (sorted unique over (image, cluster, batch * div^2 + label),
  per-image ranks, scatter into zero tables): every output identical, gradients of the prototypes too.  Views of one
  image interleaved (multiview), images without pixels, dynamic table width (M=None)."""
  import torch
  from hsg_amd.models.embeddings import hierarchy as hz
  g = torch.Generator().manual_seed(seed)
  batch = torch.sort(torch.randint(0, B, (n,), generator=g)).values
  if B > 2:
    batch = batch[batch != 1]                                   # an image without pixels
  n = batch.shape[0]
  cluster = torch.randint(0, K, (n,), generator=g)
  lab = (cluster * 7 + batch) % 5                                # one label per (batch, cluster)
  emb = torch.nn.functional.normalize(torch.randn((n, C), generator=g), dim=1)
  pos = torch.randn((n, 6), generator=g)
  image_ids = (torch.arange(B) % 2) if multiview else None      # views 0, 2, 4 .. of image 0; 1, 3 .. of image 1
  div = 16
  T = lambda t: t.to(dev) if t is not None else None
  e = T(emb).requires_grad_(True)
  out = hz.calculate_kmeans_prototypes(e, T(cluster), T(batch), T(pos), T(lab), T(image_ids), label_divisor=div,
                                       max_num_clusters=M)
  # ---- ATen restatement
  b, c, l = batch.long(), cluster.long(), lab.long()
  img = b if image_ids is None else image_ids[b]
  key2 = b * div * div + l
  trip = torch.stack([img, c, key2], 1)
  uniq, gid = torch.unique(trip, dim=0, return_inverse=True)
  P = uniq.shape[0]
  uimg = uniq[:, 0]
  imgs, img_of_seg = torch.unique(uimg, return_inverse=True)
  first = torch.ones(P, dtype=torch.bool); first[1:] = uimg[1:] != uimg[:-1]
  seg = torch.arange(P)
  local = seg - torch.cummax(torch.where(first, seg, torch.zeros_like(seg)), 0).values
  Bp, most = imgs.shape[0], int(local.max())
  Mm = M if M is not None else most + 1
  slot = img_of_seg * Mm + local
  e2 = emb.clone().requires_grad_(True)
  sums = torch.zeros((P, C)).index_add(0, gid, e2)
  protos = torch.nn.functional.normalize(sums, dim=1)
  table = torch.zeros((Bp * Mm, C)).index_copy(0, slot, protos).view(Bp, Mm, C).permute(0, 2, 1)
  cnt = torch.zeros(P).index_add(0, gid, torch.ones(n))
  pmean = torch.zeros((P, 6)).index_add(0, gid, pos) / cnt.view(-1, 1)
  ptab = torch.zeros((Bp * Mm, 6)).index_copy(0, slot, pmean).view(Bp, Mm, 6).permute(0, 2, 1)
  masks = torch.ones(Bp * Mm, dtype=torch.bool); masks[slot] = False
  plabs = torch.full((Bp * Mm,), -1, dtype=torch.long); plabs[slot] = uniq[:, 2] % (div * div)
  pbatch = torch.full((Bp * Mm,), -1, dtype=torch.long); pbatch[slot] = uniq[:, 2] // (div * div)
  by_image = local[gid]
  order = torch.argsort(img_of_seg[gid], stable=True)
  by_image = by_image[order]
  assert tuple(out[0].shape) == (Bp, C, Mm)
  assert torch.allclose(out[0].detach().cpu(), table.detach(), atol=2e-6)
  assert torch.allclose(out[1].cpu(), ptab, atol=1e-5)
  assert torch.equal(out[2].cpu(), masks.view(Bp, Mm))
  assert torch.equal(out[3].cpu(), plabs.view(Bp, Mm))
  assert torch.equal(out[4].cpu(), pbatch.view(Bp, Mm))
  assert torch.equal(out[5].cpu(), by_image)
  w = torch.randn((Bp, C, Mm), generator=g)
  (out[0] * T(w)).sum().backward()
  (table * w).sum().backward()
  assert torch.allclose(e.grad.cpu(), e2.grad, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('tlayout', ['', '0', '1'])
@pytest.mark.parametrize('shape', [(2, 256, 96, 96), (1, 256, 90, 70), (3, 256, 64, 48)])
def test_k256_tile_order_copy_straight_from_prep_vs_oracle(dev, oracle, monkeypatch, tlayout, shape):
  """K = 256 without a label map: by default the prep kernel writes the fp16 copy in tile order itself (H * W % 32 == 0)
  and the register-resident-row filter runs on it, no row-major copy, no conversion; 90 x 70 (H * W % 32 != 0) keeps
  the pair kernel; HSGK_TLAYOUT = 0 / 1: never / with the conversion kernel where prep declines.  All five outputs
  bit-exact vs the oracle, every filtered label verified on the device."""
  from hsg_amd import _lib
  from hsg_amd.utils.segsort import common as sc
  if tlayout:
    monkeypatch.setenv('HSGK_TLAYOUT', tlayout)
  else:
    monkeypatch.delenv('HSGK_TLAYOUT', raising=False)
  B, C, H, W = shape
  iters = 6
  x = synth.embeddings_nchw(synth.SEED_BASE + 41, shape, 'iid')
  loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
  _lib.verify_collect()
  _lib.verify_enable(True)
  try:
    got = _run_segkm(dev, x, None, (16, 16), None, iters)
    compared, differing = _lib.verify_collect()
  finally:
    _lib.verify_enable(False)
  assert compared == got[0].shape[0] * iters and differing == 0, (compared, differing)
  ref = oracle.segment_by_kmeans(x, None, (16, 16), loc, None, iters)
  for nm, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert a.shape == b.shape, nm
    assert np.array_equal(a, b), '%s: %d mismatching elements' % (nm, int((a != b).sum()))

@pytest.mark.gpu
@pytest.mark.parametrize('hard', ['', '0', 'skip0'])
@pytest.mark.parametrize('flavour,shape,grid', [('mixture', (2, 256, 96, 96), (16, 16)),
                                                ('mixture', (2, 384, 64, 64), (8, 16)),
                                                ('flat', (1, 256, 64, 64), (16, 16)),
                                                ('flat', (1, 130, 48, 48), (8, 12))])
def test_all_k_entries_dense_exact_pass_vs_oracle(dev, oracle, monkeypatch, hard, flavour, shape, grid):
  """K > 64 on inputs whose rows sit between MANY near-duplicate centroids (the mixture input: ~13 of the 256
  seed-grid clusters share each mixture centre; 'flat': every row within 1e-3 of one vector, all K centroids
  near-ties): their exact-queue entries ask for all K centroids and take the dense fp32 matrix-pipe pass
  (assign_hard_rows_kernel; HSGK_HARD=0: the whole-wave chains inside the exact pass).  All five outputs bit-exact
  vs the oracle on both routes, every filtered label verified on the device."""
  from hsg_amd import _lib
  from hsg_amd.utils.segsort import common as sc
  monkeypatch.delenv('HSGK_HARD', raising=False)
  monkeypatch.delenv('HSGK_HARD_SKIP', raising=False)
  if hard == 'skip0':                    # every such entry to the dense pass (default: the first few stay on the wave path)
    monkeypatch.setenv('HSGK_HARD_SKIP', '0')
  elif hard:
    monkeypatch.setenv('HSGK_HARD', hard)
  B, C, H, W = shape
  iters = 5
  if flavour == 'flat':
    v = synth.gaussish(synth.SEED_BASE + 61, C).reshape(1, C, 1, 1)
    x = (v + np.float32(1e-3) * synth.embeddings_nchw(synth.SEED_BASE + 62, shape, 'iid')).astype(np.float32)
  else:
    x = synth.embeddings_nchw(synth.SEED_BASE + 60 + C, shape, 'mixture')
  loc = (sc.generate_location_features((H, W), 'cpu', 'float') - 0.5).numpy()
  _lib.verify_collect()
  _lib.verify_enable(True)
  try:
    got = _run_segkm(dev, x, None, grid, None, iters)
    compared, differing = _lib.verify_collect()
  finally:
    _lib.verify_enable(False)
  # (C = 130: no fp16 filter for that row length -- the exact engine runs and nothing is compared)
  assert compared in ((0,) if C == 130 else ()) + (got[0].shape[0] * iters,) and differing == 0, (compared, differing)
  ref = oracle.segment_by_kmeans(x, None, grid, loc, None, iters)
  for nm, a, b in zip(('emb', 'emb_loc', 'labels', 'cluster', 'batch'), got, ref):
    assert a.shape == b.shape, nm
    assert np.array_equal(a, b), '%s: %d mismatching elements' % (nm, int((a != b).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize('case', util.F19_CASES)
def test_f19_full_size_image_vs_reference(dev, case):
  """tools/gen_golden.py f19: one WHOLE image of every BASELINE shape (448^2 / K 64, 224^2 / K 64, 768^2 / K 256,
  224^2 / C 384 / K 128; i.i.d. and mixture) against the REFERENCE's own labels.
  (1) teacher-forced, all ten iterations: segment_by_kmeans(iterations=1, cluster_indices=the reference's labels
      after iteration t - 1) == the reference's labels after iteration t, except on the recorded near-tie pixels
      (float64 margin < 1e-6 in the reference's own scores, <= 12 pixels per image and iteration);
  (2) free-running: the operator's 10 iterations end exactly where the fixture says the canonical arithmetic ends
      (the reference's labels where no near-tie flipped on the way)."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  g, x, grid, loc, ref, forced = util.f19_case(case)
  _, C, H, W = x.shape
  xd = torch.from_numpy(x).to(dev)
  ref[0] = sc.initialize_cluster_labels(list(grid), [H, W], 'cpu').view(-1).numpy()
  ref[0] = np.unique(ref[0], return_inverse=True)[1]
  for t in range(1, 11):                 # every iteration (round 6: f20 holds the reference's labels after 3 .. 8)
    init = torch.from_numpy(ref[t - 1].reshape(1, H, W)).to(dev)
    out = sc.segment_by_kmeans(xd, None, list(grid), iterations=1, cluster_indices=init)
    got = out[3].cpu().numpy()
    want = np.unique(forced(t), return_inverse=True)[1]
    assert np.array_equal(got, want), '%s iteration %d: %d pixels' % (case, t, int((got != want).sum()))
    if g['tf%d_pixels' % t].size:
      assert g['tf%d_margin64' % t].max() < util.TIE_MARGIN
  out = sc.segment_by_kmeans(xd, None, list(grid), iterations=10)
  want = ref[10].copy()
  want[g['free_pixels']] = g['free_oracle']
  got = out[3].cpu().numpy()
  want = np.unique(want, return_inverse=True)[1]
  assert np.array_equal(got, want), '%s free-running: %d pixels' % (case, int((got != want).sum()))
  # the rows themselves against the reference's (strided subset, column sums are not stored at this size)
  assert out[1].shape == (H * W, C + 2)


@pytest.mark.gpu
def test_f21_full_size_labelled_input_vs_reference(dev):
  """tools/gen_golden.py f21: the labelled path at BASELINE size against the REFERENCE's own operator (2 x 256 x 448
  x 448, 48-region label map + ignore band; common.py:355-405): segment_by_kmeans(iterations=1, cluster_indices =
  the reference's labels after nine iterations) gives the reference's kept pixels, labels, image ids, segment ids
  (except the fixture's 3 recorded near-tie pixels) and rows."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  g, x, lab, grid, loc, start, want = util.f21_case()
  out = sc.segment_by_kmeans(torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), list(grid),
                             ignore_index=255, iterations=1, cluster_indices=torch.from_numpy(start).to(dev))
  util.check_f21(g, want, *[o.cpu().numpy() for o in out])


@pytest.mark.gpu
def test_torch_extension_binding_matches_the_ctypes_mirror_bit_for_bit(dev, monkeypatch):
  """The two host bindings drive the same kernels: segment_reduce (prototypes / means / sums, forward and the
  gradient of the rows), segment_by_kmeans (five outputs and the gradient of the NCHW input, with and without a
  label map), the contrastive loss (three label sets in one pass, pixel / prototype groups; both gradients) and the
  single-rank prototype exchange (six results, gradients of both row sets) through
  torch.ops.hsgk.* (hsg_amd/csrc/torch_ops.cpp, C++ autograd nodes) and through ctypes give identical bits;
  the exchange also reproduces the reference's f8 tables and regrows its capacity the same way."""
  import torch
  from hsg_amd import _lib, _torch_ops, ops
  from hsg_amd.models import utils as mu
  monkeypatch.delenv('HSGK_BINDING', raising=False)      # (the suite may run with HSGK_BINDING=ctypes as a whole)
  assert _torch_ops.ops() is not None, 'libhsgk_torch.so is not built'
  x = torch.from_numpy(synth.gaussish(11, 5000 * 66).reshape(5000, 66).copy()).to(dev)
  lab = torch.from_numpy((synth.hash_u64(12, 5000) % np.uint64(37)).astype(np.int64)).to(dev)
  w = torch.from_numpy(synth.gaussish(13, 40 * 66).reshape(40, 66).copy()).to(dev)
  res = {}
  for binding in ('torch', 'ctypes'):
    if binding == 'ctypes':
      monkeypatch.setenv('HSGK_BINDING', 'ctypes')
    else:
      monkeypatch.delenv('HSGK_BINDING', raising=False)
    outs = []
    for mode in (0, 1, 2):
      xr = x.clone().requires_grad_(True)
      o = ops.segment_reduce(xr, lab, 40, mode)
      (o * w).sum().backward()
      outs += [o.detach().cpu().numpy(), xr.grad.cpu().numpy()]
    # segment_by_kmeans: with a label map + ignore band (one host read) and without (none), forward and backward
    from hsg_amd.utils.segsort import common as sc
    from hsg_amd.utils.segsort import loss as sl
    xs = synth.embeddings_nchw(synth.SEED_BASE + 71, (2, 64, 24, 40), 'mixture')
    lb = synth.overseg_labels(synth.SEED_BASE + 72, 2, 24, 40, regions=7, ignore_rows=2)
    for labelled in (True, False):
      xt = torch.from_numpy(xs).to(dev).requires_grad_(True)
      o = sc.segment_by_kmeans(xt, torch.from_numpy(lb).to(dev) if labelled else None, [3, 4],
                               ignore_index=255 if labelled else None, iterations=4)
      w1 = torch.from_numpy(synth.gaussish(73, o[0].numel()).reshape(o[0].shape).copy()).to(dev)
      w2 = torch.from_numpy(synth.gaussish(74, o[1].numel()).reshape(o[1].shape).copy()).to(dev)
      ((o[0] * w1).sum() + (o[1] * w2).sum()).backward()
      outs += [t.detach().cpu().numpy() for t in o] + [xt.grad.cpu().numpy()]
    # the contrastive loss: three label sets in one pass (two kappas), then one grouped set; both gradients
    n, c, P = 700, 64, 53
    e = torch.from_numpy(synth.gaussish(75, n * c).reshape(n, c).copy()).to(dev)
    e = (e / e.norm(dim=1, keepdim=True)).requires_grad_(True)
    inst = torch.from_numpy((synth.hash_u64(76, n) % np.uint64(P)).astype(np.int64)).to(dev)
    pr = ops.segment_reduce(e.detach(), inst, P, 0).requires_grad_(True)
    psems = [torch.from_numpy((synth.hash_u64(77 + i, P) % np.uint64(5 + i)).astype(np.int64)).to(dev) for i in range(3)]
    sets = [(ps[inst], ps, k, 'segsort+' if i != 1 else 'segsort') for i, (ps, k) in enumerate(zip(psems, (16.0, 16.0, 10.0)))]
    vals = sl.segsort_losses(e, inst, pr, sets)
    (vals[0] + 2 * vals[1] + 3 * vals[2]).backward()
    outs += [v.detach().cpu().numpy().reshape(1) for v in vals] + [e.grad.cpu().numpy(), pr.grad.cpu().numpy()]
    qg = torch.from_numpy((synth.hash_u64(81, n) % np.uint64(2)).astype(np.int64)).to(dev)
    pg = torch.from_numpy((np.arange(P) % 2).astype(np.int64)).to(dev)
    nl = sl.segsort_nll(e.detach(), psems[0][inst], inst, pr.detach(), psems[0], 12.0, 'segsort+', qg, pg)
    outs.append(torch.nan_to_num(nl, nan=-1.0, posinf=-2.0, neginf=-3.0).cpu().numpy())
    g = util.load('f8_exchange')
    parts = util.exchange_inputs(int(g['seed']))
    cat = {k: np.concatenate([p[k] for p in parts]) for k in ('emb', 'emb_loc', 'cluster', 'sem', 'inst')}
    # one device holding both 'GPUs' pixel sets: batch ids of the second shifted as the reference does (B * gpu)
    nb = int(parts[0]['batch'].max()) + 1
    cat['batch'] = np.concatenate([parts[0]['batch'], parts[1]['batch'] + nb * 0])
    T = lambda k: torch.from_numpy(cat[k]).to(dev)
    for cap_start in (mu._CAP_START, 8):
      mu._capacity.clear()
      saved, mu._CAP_START = mu._CAP_START, cap_start
      try:
        e, el = T('emb').requires_grad_(True), T('emb_loc').requires_grad_(True)
        r = mu.exchange_prototypes(e, el, T('cluster'), T('batch'), T('sem'), T('inst'), local=True, tag='bind_test')
        w1, w2 = util.exchange_grad_weights(0, r[0].shape[0], r[0].shape[1], r[1].shape[1])
        ((r[0] * torch.from_numpy(w1).to(dev)).sum() + (r[1] * torch.from_numpy(w2).to(dev)).sum()).backward()
        outs += [t.detach().cpu().numpy() for t in r] + [e.grad.cpu().numpy(), el.grad.cpu().numpy()]
      finally:
        mu._CAP_START = saved
    res[binding] = outs
  assert len(res['torch']) == len(res['ctypes'])
  for i, (a, b) in enumerate(zip(res['torch'], res['ctypes'])):
    assert a.shape == b.shape and a.dtype == b.dtype, i
    assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), \
        'result %d differs between the bindings' % i

