"""Multi-rank / launcher checks on real GPUs: the portable generator on the device, the
bench launcher, and the prototype exchange over RCCL (`nccl` backend) -- the world-size-2
case runs when the box shows at least two devices, the single-rank process-group case always."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from hsg_amd.utils import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FTOL = 2e-6


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def test_device_generator_matches_numpy_bits():
  import torch
  dev = torch.device('cuda:0')
  for shape in [(3, 5, 7, 9), (2, 32, 33, 17), (1, 1, 1, 1)]:
    want = synth.embeddings_nchw(synth.SEED_BASE + 9, shape, 'iid')
    got = synth.device_embeddings_nchw(synth.SEED_BASE + 9, shape, 'iid', dev).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    wantm = synth.embeddings_nchw(synth.SEED_BASE + 9, shape, 'mixture')
    gotm = synth.device_embeddings_nchw(synth.SEED_BASE + 9, shape, 'mixture', dev).cpu().numpy()
    assert np.array_equal(gotm.view(np.uint32), wantm.view(np.uint32))
  # a rank's shard = the tail of the global batch (global image index = rank * B + b)
  full = synth.embeddings_nchw(synth.SEED_BASE + 2, (5, 6, 10, 12), 'iid')
  part = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (2, 6, 10, 12), 'iid', dev, first_image=3).cpu().numpy()
  assert np.array_equal(part, full[3:])
  fullm = synth.embeddings_nchw(synth.SEED_BASE + 2, (5, 6, 10, 12), 'mixture')
  partm = synth.device_embeddings_nchw(synth.SEED_BASE + 2, (2, 6, 10, 12), 'mixture', dev, first_image=3).cpu().numpy()
  assert np.array_equal(partm, fullm[3:])


def _bench(args, env_extra=None, timeout=600):
  env = dict(os.environ)
  env.update(env_extra or {})
  env['MASTER_PORT'] = str(_free_port())
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT,
                     capture_output=True, text=True, timeout=timeout)
  assert p.returncode == 0, p.stderr[-2000:]
  lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, p.stdout[-2000:]
  return json.loads(lines[0])


def test_bench_line_single_rank_process_group():
  """bench.py with an RCCL process group of one rank: contract fields, the exchange field."""
  r = _bench(['--workload', 'cfg1', '--steps', '2', '--warmup', '1', '--cpu-images', '0', '--no-extra'],
             {'HSGK_BENCH_FORCE_DIST': '1'})
  assert r['n_gpus'] == 1 and r['steps'] == 2 and r['scaling'] == 'weak'
  assert r['value'] > 0 and (r['roofline']['frac'] is None or r['roofline']['frac'] <= 1.0)   # (no counter pass for cfg1)
  assert 'error' not in r['prototype_exchange'], r['prototype_exchange']
  assert r['exchange_ms'] > 0


def _dist_worker(rank, world, port, result_dir, backend, same_device, library_comm):
  import torch
  import torch.distributed as dist
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  index = 0 if same_device else rank
  torch.cuda.set_device(index)
  dev = torch.device('cuda', index)
  kw = {'device_id': dev} if backend == 'nccl' else {}
  dist.init_process_group(backend, rank=rank, world_size=world, **kw)
  try:
    from hsg_amd.models import utils as mu
    mu.use_library_comm = library_comm
    g = util.load('f8_exchange')
    part = util.exchange_inputs(int(g['seed']))[rank]
    T = lambda k: torch.from_numpy(part[k]).to(dev)
    emb = T('emb').requires_grad_(True)
    emb_loc = T('emb_loc').requires_grad_(True)
    before = mu.collective_calls
    protos, protos_loc, psem, pinst, pbatch, upd = mu.gather_clustering_and_update_prototypes(
        emb, emb_loc, T('cluster'), T('batch'), T('sem'), T('inst'))
    assert mu.collective_calls - before == 2
    assert np.array_equal(psem.cpu().numpy(), g['psem'])
    assert np.array_equal(pinst.cpu().numpy(), g['pinst'])
    assert np.array_equal(pbatch.cpu().numpy(), g['pbatch'])
    assert np.array_equal(upd.cpu().numpy(), g['upd%d' % rank])
    assert np.abs(protos.detach().cpu().numpy() - g['protos']).max() <= FTOL
    assert np.abs(protos_loc.detach().cpu().numpy() - g['protos_loc']).max() <= FTOL
    # the oracle's restatement (per-source C2 sums, added in source order) bit for bit
    from oracle import oracle as orc
    want = orc.exchange_prototypes(util.exchange_inputs(int(g['seed'])))
    assert np.array_equal(protos.detach().cpu().numpy().view(np.uint32), want[0].view(np.uint32))
    assert np.array_equal(protos_loc.detach().cpu().numpy().view(np.uint32), want[1].view(np.uint32))
    # the gradient that crosses the collective, against the reference's own autograd (f8: gemb / gloc)
    util.check_exchange_grads(g, rank, protos, protos_loc, emb, emb_loc)
    # too small tuple blocks: every rank regrows from the gathered header counts and repeats
    mu._capacity.clear()
    saved, mu._CAP_START = mu._CAP_START, 8
    try:
      again = mu.gather_clustering_and_update_prototypes(
          T('emb'), T('emb_loc'), T('cluster'), T('batch'), T('sem'), T('inst'))
      assert np.array_equal(again[5].cpu().numpy(), g['upd%d' % rank])
      assert np.array_equal(again[0].cpu().numpy().view(np.uint32), want[0].view(np.uint32))
    finally:
      mu._CAP_START = saved
    mu.use_library_comm = False
    img = mu.gather_and_reorder_image_indices(T('image_id'))
    assert np.array_equal(img.cpu().numpy(), g['img%d' % rank])
    mapping = mu.gather_and_update_cluster_mappings(upd, T('cluster'))
    assert np.array_equal(mapping.cpu().numpy(), g['mapping'])
    open(os.path.join(result_dir, 'ok%d' % rank), 'w').write('ok')
  finally:
    dist.destroy_process_group()


def test_exchange_world2_on_one_device(tmp_path):
  """Two ranks sharing the box's ONE GPU: the libhsgk phases (keys -> merge over two gathered blocks ->
  sums into the global table rows -> finish) with world = 2 on hardware, the two collectives over gloo on
  device tensors (RCCL refuses several ranks per device: profiles/r03_rccl_same_device_probe.txt), against
  the reference's golden outputs and bit for bit against the oracle."""
  import torch.multiprocessing as mp
  mp.spawn(_dist_worker, args=(2, _free_port(), str(tmp_path), 'gloo', True, False), nprocs=2, join=True)
  assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()


@pytest.mark.parametrize('library_comm', [False, True])
def test_exchange_world2_rccl(tmp_path, library_comm):
  """The same over RCCL between two GPUs: through the process group, and in-stream on libhsgk's own
  communicator (hsgk_comm_*: unique id from rank 0, carried by the process group)."""
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip('true peer-to-peer: needs two GPUs (the 8-GPU node of the driver)')
  import torch.multiprocessing as mp
  mp.spawn(_dist_worker, args=(2, _free_port(), str(tmp_path), 'nccl', False, library_comm), nprocs=2, join=True)
  assert (tmp_path / 'ok0').exists() and (tmp_path / 'ok1').exists()


def test_bench_dry_ranks_on_one_device():
  """bench.py --dry-ranks 2: two ranks on the one GPU (gloo transport), every rank its own shard of the
  global batch (B * rank image offsets), max-over-ranks timing, the exchange with world = 2."""
  r = _bench(['--dry-ranks', '2', '--workload', 'cfg1', '--steps', '2', '--warmup', '1', '--cpu-images', '0'])
  assert r['dry_ranks'] == 2 and r['dist_backend'] == 'gloo' and r['rccl_ranks'] == 0
  assert r['config']['global_batch'] == 8 and r['config']['parallelism'] == 'dp2'
  px = r['prototype_exchange']
  assert 'error' not in px, px
  assert px['collectives_per_call'] == 2 and px['segments_total'] == 2 * 4 * 8       # 2 ranks x 4 images x 8 clusters
  one = _bench(['--workload', 'cfg1', '--steps', '2', '--warmup', '1', '--cpu-images', '0', '--no-extra'])
  assert one['prototype_exchange']['segments_total'] == 4 * 8


def test_bench_spawns_its_ranks():
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip('true peer-to-peer: needs two GPUs')
  r = _bench(['--gpus', '2', '--workload', 'cfg3', '--steps', '2', '--warmup', '1', '--cpu-images', '0'])
  assert r['n_gpus'] == 2 and r['config']['global_batch'] == 32 and r['rccl_ranks'] == 2
  assert 'error' not in r['prototype_exchange'] and r['prototype_exchange']['collectives_per_call'] == 2
  assert 'error' not in r['prototype_exchange_instream'], r['prototype_exchange_instream']


def test_torch_free_cpp_host_of_the_c_abi():
  """examples/cabi_host.cpp: a C++ program that links libhsgk.so and nothing of torch -- host helpers for
  the seed map / location features, hipMalloc'd buffers, its own stream -- runs segment_by_kmeans twice
  and checks determinism, unit rows and dense ids itself (exit code 0)."""
  exe = os.path.join(ROOT, 'examples', 'cabi_host')
  assert os.path.exists(exe), 'examples/cabi_host is built by __graft_entry__.build()'
  for argv in (['4', '256', '96', '80', '8', '8', '10'], ['3', '128', '28', '28', '4', '4', '5'],
               ['2', '64', '33', '47', '5', '7', '3']):
    p = subprocess.run([exe] + argv, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-500:])
    assert '-> OK' in p.stdout
