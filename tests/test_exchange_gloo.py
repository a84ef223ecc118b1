"""Multi-GPU path on CPU: world_size-2 `gloo` run of the prototype exchange
(hsg_amd/models/utils.py) against the reference's own outputs for the same
two-GPU inputs (tests/golden/f8_exchange.npz).  The local segment sums and the
normalisation -- libhsgk kernels in production -- are replaced by the CPU
oracle through the module's hooks, so the test exercises exactly the
collective / bookkeeping logic that runs over RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _install_cpu_hooks(mu):
  """The per-device kernels (libhsgk in production) are replaced by the oracle-backed backend of
  tests/exchange_cpu_backend.py; the orchestration under test stays hsg_amd/models/utils.py."""
  from tests.exchange_cpu_backend import CpuExchangeBackend
  mu.backend_class = CpuExchangeBackend


def _worker(rank, world, port, result_dir):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from hsg_amd.models import utils as mu
    _install_cpu_hooks(mu)
    g = util.load('f8_exchange')
    part = util.exchange_inputs(int(g['seed']))[rank]
    T = lambda k: torch.from_numpy(part[k])
    emb = T('emb').requires_grad_(True)
    emb_loc = T('emb_loc').requires_grad_(True)
    before = mu.collective_calls
    protos, protos_loc, psem, pinst, pbatch, upd = mu.gather_clustering_and_update_prototypes(
        emb, emb_loc, T('cluster'), T('batch'), T('sem'), T('inst'))
    assert mu.collective_calls - before == 2, 'the exchange is one all_gather + one all_reduce'
    assert np.array_equal(psem.numpy(), g['psem'])
    assert np.array_equal(pinst.numpy(), g['pinst'])
    assert np.array_equal(pbatch.numpy(), g['pbatch'])
    assert np.array_equal(upd.numpy(), g['upd%d' % rank])
    assert np.abs(protos.detach().numpy() - g['protos']).max() <= 2e-6
    assert np.abs(protos_loc.detach().numpy() - g['protos_loc']).max() <= 2e-6
    # gradient crosses the collective: every rank's loss sees the whole table
    # (every rank backpropagates ITS scalar; the reference's gradients are the sum over both replicas)
    before = mu.collective_calls
    util.check_exchange_grads(g, rank, protos, protos_loc, emb, emb_loc)
    assert mu.collective_calls - before == 1, 'the backward is one all_reduce'

    img = mu.gather_and_reorder_image_indices(T('image_id'))
    assert np.array_equal(img.numpy(), g['img%d' % rank])
    before = mu.collective_calls
    mapping = mu.gather_and_update_cluster_mappings(upd, T('cluster'))
    assert mu.collective_calls - before == 1
    # the reference's duplicate-index assignment on CPU writes in ascending key order, the
    # last (largest) partner wins: the golden vector is compared exactly
    assert np.array_equal(mapping.numpy(), g['mapping'])
    # a rank without rows still takes part in the collectives
    empty = torch.zeros((0,), dtype=torch.long)
    m2 = mu.gather_and_update_cluster_mappings(upd if rank == 0 else empty,
                                               T('cluster') if rank == 0 else empty)
    assert m2.shape[0] == int(g['upd0'].max()) + 1
    # the tuple blocks of the prototype exchange overflow too: 4 rows of capacity against 40 - 70 segments
    # per rank -> every rank regrows from the gathered header counts and repeats; results unchanged
    mu._capacity.clear()
    saved_cap, mu._CAP_START = mu._CAP_START, 4
    try:
      before = mu.collective_calls
      again = mu.gather_clustering_and_update_prototypes(
          T('emb'), T('emb_loc'), T('cluster'), T('batch'), T('sem'), T('inst'))
      assert mu.collective_calls - before > 2
      assert np.array_equal(again[5].numpy(), g['upd%d' % rank]) and np.array_equal(again[2].numpy(), g['psem'])
      assert np.abs(again[0].numpy() - g['protos']).max() <= 2e-6
      assert mu._cap_get(None, 'proto') >= int(g['psem'].shape[0]) // 2
    finally:
      mu._CAP_START = saved_cap
    # capacity overflow: every rank regrows from the same counts and repeats the gather
    mu._capacity.clear()
    saved_cap, mu._CAP_START = mu._CAP_START, 4
    try:
      rows = torch.arange(10 + 7 * rank, dtype=torch.long).view(-1, 1) + 100 * rank
      got, counts = mu._all_gather_rows(rows, None, 'test_overflow')
      assert counts == [10, 17]
      want = torch.cat([torch.arange(10).view(-1, 1), torch.arange(17).view(-1, 1) + 100], 0)
      assert torch.equal(got, want)
      f = torch.arange(6, dtype=torch.float32).view(3, 2) * (rank + 1)
      gotf, _ = mu._all_gather_rows(f[:2 + rank], None, 'test_float')
      assert torch.equal(gotf, torch.cat([f[:2] / (rank + 1), f[:3] / (rank + 1) * 2], 0))
    finally:
      mu._CAP_START = saved_cap
    # one process per GPU: the image offset of segment_by_kmeans follows the global rank, not
    # the local device index (every isolated rank sees device 0)
    from hsg_amd.utils.segsort import common as sc
    assert sc._batch_offset(6, torch.device('cpu')) == 6 * rank
    assert sc._batch_offset(6, torch.device('cuda', 0)) == 6 * rank
    datas = mu.gather_and_update_datas(T('emb')[:5])
    assert np.array_equal(datas.numpy(), g['datas'])
    # list form (one tensor per GPU of this process) returns lists
    outs = mu.gather_and_update_datas([T('emb')[:5]])
    assert isinstance(outs, list) and np.array_equal(outs[0].numpy(), g['datas'])
    open(os.path.join(result_dir, 'ok%d' % rank), 'w').write('ok')
  finally:
    dist.destroy_process_group()


def test_exchange_world2_gloo(tmp_path):
  port = _free_port()
  mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()


def test_exchange_single_process_list_api(oracle):
  """Reference calling convention (lists, one tensor per GPU of ONE process, train.py:190) without
  torch.distributed: keys and sums per 'device', only the tuple blocks and the sum tables are joined on
  the anchor (no pixel row is moved: the backend objects never see another device's rows)."""
  from hsg_amd.models import utils as mu
  saved = mu.backend_class
  _install_cpu_hooks(mu)
  try:
    g = util.load('f8_exchange')
    parts = util.exchange_inputs(int(g['seed']))
    T = lambda k: [torch.from_numpy(p[k]) for p in parts]
    protos, protos_loc, psem, pinst, pbatch, upd = mu.gather_clustering_and_update_prototypes(
        T('emb'), T('emb_loc'), T('cluster'), T('batch'), T('sem'), T('inst'), 'cpu')
    assert isinstance(protos, list) and len(protos) == 2
    assert np.array_equal(psem[0].numpy(), g['psem'])
    assert np.array_equal(pinst[1].numpy(), g['pinst'])
    assert np.array_equal(pbatch[0].numpy(), g['pbatch'])
    assert np.array_equal(upd[0].numpy(), g['upd0']) and np.array_equal(upd[1].numpy(), g['upd1'])
    assert np.abs(protos[0].numpy() - g['protos']).max() <= 2e-6
    assert np.abs(protos_loc[1].numpy() - g['protos_loc']).max() <= 2e-6
    img = mu.gather_and_reorder_image_indices(T('image_id'), 'cpu')
    assert np.array_equal(img[0].numpy(), g['img0']) and np.array_equal(img[1].numpy(), g['img1'])
    mapping = mu.gather_and_update_cluster_mappings(upd, T('cluster'), 'cpu')
    uniq_pairs = {}
    for a, b in zip(np.concatenate([g['upd0'], g['upd1']]), np.concatenate([p['cluster'] for p in parts])):
      uniq_pairs.setdefault(int(a), set()).add(int(b))
    for a, bs in uniq_pairs.items():
      assert int(mapping[0][a]) == max(bs) == int(g['mapping'][a])
    assert np.array_equal(mapping[0].numpy(), g['mapping'])
  finally:
    mu.backend_class = saved
