"""CPU-only checks: the C-ABI library loads, exports every symbol that
include/hsgk.h declares (no compute calls without a GPU), and the host-side
helpers of the Python mirror agree with the oracle / golden tables."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest

from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  text = open(os.path.join(ROOT, 'include', 'hsgk.h')).read()
  return sorted(set(re.findall(r'HSGK_API\s+[\w\s\*]+?\b(hsgk_\w+)\s*\(', text)))


def test_header_symbols_exported_and_bound():
  from hsg_amd import _lib
  L = _lib.lib()
  names = _declared_symbols()
  assert len(names) >= 10
  for n in names:
    assert hasattr(L, n), 'libhsgk.so does not export %s' % n
    assert n in _lib.SIGNATURES, 'hsg_amd/_lib.py does not bind %s' % n
  assert L.hsgk_version() >= 100
  assert set(_lib.SIGNATURES) == set(names)


def test_torch_extension_binding_loads_and_registers_its_ops():
  """hsg_amd/csrc/torch_ops.cpp (TORCH_LIBRARY(hsgk, ...) + C++ autograd nodes): the library loads next
  to libhsgk.so, binds the same ABI version, registers the op schemas -- and refuses CPU tensors (no compute here)."""
  import torch
  from hsg_amd import _lib, _torch_ops
  tops = _torch_ops.ops()
  assert tops is not None, 'libhsgk_torch.so is not built (make -C hsg_amd/csrc torch)'
  assert int(tops.abi_version()) == _lib.ABI_VERSION
  for name in ('segment_reduce', 'exchange_local', 'segsort_nll', 'segment_by_kmeans'):
    schema = str(getattr(tops, name).default._schema)
    assert schema.startswith('hsgk::' + name), schema
  with pytest.raises(RuntimeError, match='no CPU path'):
    tops.segment_reduce(torch.zeros(3, 4), torch.zeros(3, dtype=torch.long), 2, 0)


def test_workspace_queries_run_on_cpu():
  from hsg_amd import _lib
  L = _lib.lib()
  a = L.hsgk_segment_by_kmeans_workspace_bytes(4, 32, 64, 64, 8, 32)
  b = L.hsgk_segment_by_kmeans_workspace_bytes(48, 256, 448, 448, 64, 48 * 64)
  assert 0 < a < b < (1 << 33)        # cfg2: 5.6 GB, mostly the fp16 copy of the rows
  assert L.hsgk_kmeans_workspace_bytes(5000, 258, 64) > 0
  assert L.hsgk_lloyd_workspace_bytes(2, 4096, 34, 8) > 0


def test_missing_library_fails_loudly(monkeypatch):
  from hsg_amd import _lib
  monkeypatch.setattr(_lib, '_lib', None)
  monkeypatch.setattr(_lib, 'SO_PATH', '/nonexistent/libhsgk.so')
  with pytest.raises(_lib.HsgkError):
    _lib.lib()


def test_cpu_tensors_rejected():
  import torch
  from hsg_amd import _lib
  from hsg_amd.utils.segsort import common as sc
  with pytest.raises(_lib.HsgkError):
    sc.segment_by_kmeans(torch.zeros(1, 4, 8, 8), None, [2, 2])


def test_seed_axis_matches_torch_and_oracle(oracle):
  """Host seed labels (torch linspace().round_()) == the oracle's restatement of ATen's
  float32 linspace, for every grid size up to 32 and every image side up to 700 (the exactly
  rounded i*(k-1)/(n-1) differs for about 2 % of these pairs, e.g. k = 4, n = 43)."""
  import torch
  from hsg_amd.utils.segsort import common as sc
  for n in list(range(1, 700)) + [768, 1024, 2048]:
    for k in range(1, 33):
      t = torch.linspace(0, k - 1, n).round_().long().numpy()
      assert np.array_equal(t, oracle.grid_seed_axis(k, n)), (n, k)
    assert np.array_equal(torch.linspace(0, 1, n).numpy(), oracle.linspace_f32(0.0, 1.0, n)), n
  assert np.array_equal(sc.generate_location_features((37, 53), 'cpu', 'float').numpy(),
                        oracle.generate_location_features((37, 53)))
  lab = sc.initialize_cluster_labels([8, 8], (448, 448), 'cpu').numpy()
  assert np.array_equal(lab, oracle.initialize_cluster_labels((8, 8), (448, 448)))
  assert np.bincount(lab.reshape(-1)).min() == 1024          # edge cells half width


def test_default_location_features_match_golden():
  from hsg_amd.utils.segsort import common as sc
  g = util.load('f2_linspace01')
  for key in ('n56', 'n448', 'n768'):
    n = int(key[1:])
    loc = sc.generate_location_features((n, 3), 'cpu', 'float').numpy()
    assert np.array_equal(loc[:, 0, 0], g[key])


def test_host_helpers_match_torch_tables():
  """The library's own host helpers (for non-Python hosts of the C ABI) give the seed map and the
  location features the torch-based mirror computes -- the float32 linspace bits included."""
  import ctypes
  import numpy as np
  import torch
  from hsg_amd import _lib
  from hsg_amd.utils.segsort import common as sc
  L = _lib.lib()
  for (ky, kx, H, W) in [(8, 8, 448, 448), (4, 4, 43, 28), (2, 4, 64, 64), (16, 16, 768, 768), (8, 16, 224, 224),
                         (6, 6, 28, 28), (5, 7, 33, 47), (1, 1, 9, 5), (12, 24, 128, 256), (3, 9, 700, 31)]:
    seed = np.empty((H * W,), np.int32)
    K = ctypes.c_int32(0)
    assert L.hsgk_host_grid_seed_map(ky, kx, H, W, seed.ctypes.data_as(ctypes.c_void_p), ctypes.byref(K)) == 0
    want, wantK = sc._seed_map([ky, kx], H, W, torch.device('cpu'))
    assert K.value == wantK and np.array_equal(seed, want.numpy()), (ky, kx, H, W)
    loc = np.empty((H, W, 2), np.float32)
    assert L.hsgk_host_location_features(H, W, loc.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(loc.view(np.uint32), sc._default_loc(H, W, torch.device('cpu')).numpy().view(np.uint32))


def test_one_hot_helper_matches_the_reference_definition():
  """general/common.py:76-98: an (N+1)-D long tensor with a single 1 per label, width max + 1 by default."""
  import torch
  from hsg_amd.utils.general import common as gc
  lab = torch.tensor([[0, 3, 1], [2, 2, 0]])
  got = gc.one_hot(lab)
  assert got.dtype == torch.long and tuple(got.shape) == (2, 3, 4)
  assert torch.equal(got, torch.nn.functional.one_hot(lab, 4))
  assert tuple(gc.one_hot(lab, 7).shape) == (2, 3, 7)


def test_bookkeeping_helpers_of_the_swapped_modules():
  """resize_labels / pca / calculate_principal_components (general/common.py:11-73) and get_params
  (models/utils.py:12-38): ATen helpers outside the hot path, present so that the import swap is complete."""
  import torch
  from hsg_amd.utils.general import common as gc
  from hsg_amd.models import utils as mu
  lab = torch.arange(2 * 6 * 8).reshape(2, 6, 8)
  small = gc.resize_labels(lab, (3, 4))
  assert small.dtype == torch.long and tuple(small.shape) == (2, 3, 4)
  assert torch.equal(small, lab[:, ::2, ::2])
  x = torch.randn(50, 7, generator=torch.Generator().manual_seed(3))
  pc = gc.calculate_principal_components(x, 3)
  assert tuple(pc.shape) == (7, 3) and torch.allclose(pc.t() @ pc, torch.eye(3), atol=1e-5)
  y = gc.pca(x.reshape(5, 10, 7), 3)
  assert tuple(y.shape) == (5, 10, 3) and torch.allclose(y.reshape(50, 3), x @ pc, atol=1e-5)
  # the variance captured decreases along the components
  v = (gc.pca(x - x.mean(0, keepdim=True), 3) ** 2).sum(0)
  assert v[0] >= v[1] >= v[2]
  net = torch.nn.Sequential()
  net.add_module('head', torch.nn.Linear(3, 2))
  net.add_module('body', torch.nn.Linear(3, 2))
  got = list(mu.get_params(net, ['head'], ['weight']))
  assert len(got) == 1 and got[0] is net.head.weight
  assert [p is net.head.bias for p in mu.get_params(net, ['head', 'body'], ['bias'], exclude='body')] == [True]


def test_notes_ride_on_the_tensor_object_and_die_with_in_place_writes():
  """`ops.note` / `ops.noted` (host-side facts remembered on a tensor so that the next function needs no device read):
  valid for that object and version only."""
  import torch
  from hsg_amd import ops
  if not ops._notes_on:
    pytest.skip('HSGK_NO_NOTES is set')
  t = torch.arange(6)
  assert ops.noted(t, 'index_count') is None
  ops.note(t, 'index_count', 6)
  assert ops.noted(t, 'index_count') == 6
  assert ops.noted(t.long() if t.dtype != torch.int64 else t, 'index_count') == 6      # (.long() of an int64 tensor is the object itself)
  assert ops.noted(t + 0, 'index_count') is None                                       # a new tensor carries nothing
  t[0] = 5                                                                               # in-place write: the note is stale
  assert ops.noted(t, 'index_count') is None
  assert ops.noted(None, 'index_count') is None


def test_process_group_key_is_the_c10d_name_not_the_object_id():
  from hsg_amd.models import utils as mu

  class Named:
    group_name = '7'

  class Anonymous:
    pass
  assert mu._group_key(None) is None
  assert mu._group_key(Named()) == ('name', '7') == mu._group_key(Named())              # two objects, one group
  a = Anonymous()
  assert mu._group_key(a) == ('id', id(a))


def test_bench_roofline_byte_sources_come_from_the_newest_counter_pass():
  """bench.py prices each workload's launch groups with the moved bytes of that workload's newest committed
  rocprofv3 counter pass, kernels found by name prefix (no stale kernel lists, no fraction above 1
  from an algorithmic stand-in)."""
  import importlib.util
  spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  argv, sys.argv = sys.argv, ['bench.py']
  try:
    spec.loader.exec_module(bench)
  finally:
    sys.argv = argv
  rows = {'cfg2': 48 * 448 * 448, 'cfg3': 16 * 224 * 224, 'cfg4': 4 * 768 * 768, 'cfg5': 24 * 224 * 224}
  for wl, n in rows.items():
    for group in ('assign', 'accumulate', 'prep'):
      traffic, src, names, commit = bench.pmc_traffic(group, wl)
      newest = next(f for f in bench.pmc_files(wl) if os.path.exists(os.path.join(ROOT, 'profiles', f)))
      assert traffic and src == 'profiles/' + newest and newest >= 'r06_', (wl, group, src)
      assert commit, 'the counter file names the commit it was taken at'
      assert all(any(pre in k for pre in bench.PMC_GROUPS[group]) for k in names)
      assert 50 <= traffic / n <= 6000, (wl, group, traffic / n)      # bytes per pixel row and group instance (prep at C = 384: 5.4 KB)
    d = 258 if wl != 'cfg5' else 386
    e_bytes = bench.pmc_traffic('assign', wl)[0] / n
    assert e_bytes < 4 * d + 8, 'the filtered E-step moves less than one fp32 read of every row'
  assert bench.pmc_traffic('assign', 'cfg1') == (None, None, None, None)
