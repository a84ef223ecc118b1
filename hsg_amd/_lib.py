"""ctypes binding of libhsgk.so (include/hsgk.h).

The library is the product: if it is missing or fails to load, importing the
operators fails loudly -- there is no CPU / eager fallback.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, 'csrc', 'libhsgk.so')

CHUNK = 2048
EPS = 1e-12


class HsgkError(RuntimeError):
  pass


class SegkmMeta(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int64) for n in (
      'n_rows', 'n_segments', 'label_min', 'label_max', 'n_chunks', 'error',
      'relabel_mode', 'relabel_L')]


class SegkmArgs(ctypes.Structure):
  _fields_ = [
      ('embeddings', ctypes.c_void_p), ('labels', ctypes.c_void_p),
      ('loc', ctypes.c_void_p), ('loc_batch_stride', ctypes.c_int64),
      ('seed_map', ctypes.c_void_p),
      ('B', ctypes.c_int32), ('C', ctypes.c_int32), ('H', ctypes.c_int32),
      ('W', ctypes.c_int32), ('K', ctypes.c_int32), ('iterations', ctypes.c_int32),
      ('has_ignore', ctypes.c_int32),
      ('ignore_index', ctypes.c_int64), ('batch_offset', ctypes.c_int64),
      ('table_cap', ctypes.c_int64),
      ('out_embeddings', ctypes.c_void_p), ('out_embeddings_loc', ctypes.c_void_p),
      ('out_labels', ctypes.c_void_p), ('out_cluster', ctypes.c_void_p),
      ('out_batch', ctypes.c_void_p), ('meta', ctypes.c_void_p),
      ('out_norms', ctypes.c_void_p), ('out_rowmap', ctypes.c_void_p),
      ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t),
      ('seed_batch_stride', ctypes.c_int64), ('flags', ctypes.c_int32)]


class ExchangeArgs(ctypes.Structure):
  _fields_ = [
      ('embeddings', ctypes.c_void_p), ('embeddings_loc', ctypes.c_void_p),
      ('cluster', ctypes.c_void_p), ('batch', ctypes.c_void_p), ('semantic', ctypes.c_void_p),
      ('instance', ctypes.c_void_p),
      ('n', ctypes.c_int64), ('C', ctypes.c_int32), ('D', ctypes.c_int32),
      ('cap_local', ctypes.c_int64), ('cap_total', ctypes.c_int64), ('pool_rows', ctypes.c_int64),
      ('eps', ctypes.c_float),
      ('table', ctypes.c_void_p), ('prototypes', ctypes.c_void_p), ('prototypes_loc', ctypes.c_void_p),
      ('norms', ctypes.c_void_p),
      ('proto_semantic', ctypes.c_void_p), ('proto_instance', ctypes.c_void_p), ('proto_batch', ctypes.c_void_p),
      ('updated_cluster', ctypes.c_void_p), ('meta', ctypes.c_void_p),
      ('workspace', ctypes.c_void_p), ('workspace_bytes', ctypes.c_size_t)]


class LossSet(ctypes.Structure):
  _fields_ = [('sem', ctypes.c_void_p), ('psem', ctypes.c_void_p), ('kappa', ctypes.c_float),
              ('mode', ctypes.c_int32)]


_lib = None
ABI_VERSION = 401          # HSGK_VERSION the struct layouts / signatures below were written for

_vp, _i64, _i32, _f32, _sz = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                              ctypes.c_float, ctypes.c_size_t)

# name -> (restype, argtypes); must list every symbol include/hsgk.h declares
SIGNATURES = {
    'hsgk_version': (_i32, []),
    'hsgk_last_error': (ctypes.c_char_p, []),
    'hsgk_small_map_groups': (_i32, [_i32, _i32, _i32, _i32, _i32]),
    'hsgk_host_grid_seed_map': (_i32, [_i32, _i32, _i32, _i32, _vp, _vp]),
    'hsgk_host_location_features': (_i32, [_i32, _i32, _vp]),
    'hsgk_normalize_rows': (_i32, [_vp, _i64, _i32, _f32, _vp, _vp, _vp]),
    'hsgk_segment_by_kmeans_workspace_bytes': (_sz, [_i32, _i32, _i32, _i32, _i32, _i64]),
    'hsgk_segment_by_kmeans': (_i32, [ctypes.POINTER(SegkmArgs), _vp]),
    'hsgk_segment_by_kmeans_bwd': (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32,
                                          _vp, _vp]),
    'hsgk_kmeans_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'hsgk_kmeans_with_initial_labels': (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _vp, _sz, _vp]),
    'hsgk_profile_enable': (None, [_i32]),
    'hsgk_profile_collect': (_i32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]),
    'hsgk_verify_enable': (None, [_i32]),
    'hsgk_verify_collect': (_i32, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    'hsgk_lloyd_workspace_bytes': (_sz, [_i32, _i64, _i32, _i32]),
    'hsgk_lloyd_mstep': (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    'hsgk_lloyd_mstep_exact': (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'hsgk_lloyd_estep': (_i32, [_vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _sz, _vp]),
    'hsgk_segment_reduce_workspace_bytes': (_sz, [_i64, _i32, _i64]),
    'hsgk_segment_reduce': (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _f32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'hsgk_segment_reduce_bwd': (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _f32, _vp, _vp, _vp]),
    'hsgk_segsort_loss_workspace_bytes': (_sz, [_i64, _i32, _i64, _i32]),
    'hsgk_segsort_loss_fwd': (_i32, [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _sz, _vp]),
    'hsgk_segsort_loss_bwd_workspace_bytes': (_sz, [_i64, _i32, _i64, _i32]),
    'hsgk_segsort_loss_bwd': (_i32, [_vp, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _sz, _vp]),
    'hsgk_lloyd_requeued_rows': (_i32, [_i32, _i64, _i32, _i32, _vp, _sz, _vp, _vp]),
    'hsgk_hier_assign': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'hsgk_hier_assign_bwd': (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'hsgk_group_mean': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    'hsgk_group_mean_bwd': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    'hsgk_gather_labels': (_i32, [_vp, _i32, _vp, _vp, _i64, _vp, _vp]),
    'hsgk_pad_prototype_tables': (_i32, [_vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hsgk_cluster_topk': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    'hsgk_overlap_accumulate': (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    'hsgk_overlap_finish': (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    'hsgk_majority_labels': (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp]),
    'hsgk_knn_affinity': (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    'hsgk_dmon_pool_workspace_bytes': (_sz, [_i32, _i32, _i32]),
    'hsgk_dmon_pool_fwd': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    'hsgk_dmon_pool_bwd': (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    'hsgk_topk_workspace_bytes': (_sz, [_i64, _i32, _i64, _i32]),
    'hsgk_topk_prototypes': (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _sz, _vp]),
    'hsgk_topk_prototypes_grouped': (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'hsgk_synth_gaussish': (_i32, [ctypes.c_uint64, ctypes.c_uint64, _i64, _vp, _vp]),
    'hsgk_synth_mixture': (_i32, [ctypes.c_uint64, ctypes.c_uint64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    'hsgk_assign_workspace_bytes': (_sz, [_i64, _i32, _i32]),
    'hsgk_exchange_workspace_bytes': (_sz, [_i64, _i32, _i32, _i64, _i64, _i32, _i64]),
    'hsgk_exchange_prototypes': (_i32, [ctypes.POINTER(ExchangeArgs), _vp, _i32, _i32, _vp]),
    'hsgk_exchange_begin': (_i32, [ctypes.POINTER(ExchangeArgs), _vp, _i32, _i32, _vp]),
    'hsgk_exchange_finish': (_i32, [ctypes.POINTER(ExchangeArgs), _i64, _vp, _i32, _vp]),
    'hsgk_exchange_keys': (_i32, [ctypes.POINTER(ExchangeArgs), _i32, _vp]),
    'hsgk_exchange_send_block': (_vp, [ctypes.POINTER(ExchangeArgs), _i32, ctypes.POINTER(ctypes.c_size_t)]),
    'hsgk_exchange_recv_blocks': (_vp, [ctypes.POINTER(ExchangeArgs), _i32]),
    'hsgk_exchange_merge': (_i32, [ctypes.POINTER(ExchangeArgs), _i32, _i32, _vp, _vp]),
    'hsgk_exchange_sums': (_i32, [ctypes.POINTER(ExchangeArgs), _i32, _i32, _vp, _vp]),
    'hsgk_comm_unique_id': (_i32, [_vp, _sz]),
    'hsgk_comm_init_rank': (_i32, [ctypes.POINTER(ctypes.c_void_p), _i32, _i32, _vp, _sz]),
    'hsgk_comm_destroy': (_i32, [_vp]),
    'hsgk_comm_all_reduce_f32': (_i32, [_vp, _i64, _vp, _vp]),
    'hsgk_comm_all_gather_bytes': (_i32, [_vp, _vp, _sz, _vp, _vp]),
    'hsgk_find_nearest_prototypes': (_i32, [_vp, _i64, _i32, _vp, _i32, _vp, _vp, _sz, _vp]),
}


def lib():
  """Loads libhsgk.so once; raises HsgkError if it is not built."""
  global _lib
  if _lib is None:
    if not os.path.exists(SO_PATH):
      raise HsgkError(
          'libhsgk.so is not built (%s). Run `python -c "import __graft_entry__ as g; '
          'g.build()"` or `make -C hsg_amd/csrc`. There is no fallback path.' % SO_PATH)
    # torch ships its own libamdhip64; load it FIRST so libhsgk binds to the
    # same HIP runtime instance (streams and device pointers are then shared).
    import torch  # noqa: F401
    try:
      L = ctypes.CDLL(SO_PATH)
    except OSError as e:
      raise HsgkError('cannot load %s: %s' % (SO_PATH, e))
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(L, name)
      fn.restype = res
      fn.argtypes = args
    if L.hsgk_version() != ABI_VERSION:
      raise HsgkError('%s is version %d but hsg_amd/_lib.py binds version %d: rebuild it (`make -C hsg_amd/csrc`)'
                      % (SO_PATH, L.hsgk_version(), ABI_VERSION))
    _lib = L
  return _lib


def check(rc):
  if rc != 0:
    raise HsgkError('libhsgk error %d: %s' % (rc, lib().hsgk_last_error().decode()))
  if _deferred:                            # (cheap: a list check; completed flags are looked at by every libhsgk call)
    poll_deferred()


PROF_KINDS = ('prep', 'accumulate', 'finalize', 'assign', 'relabel')


def profile_enable(on):
  lib().hsgk_profile_enable(int(bool(on)))


def profile_collect():
  """{kind: (total_ms, launches)} since the last collect; waits for the events."""
  ms = (ctypes.c_double * len(PROF_KINDS))()
  cnt = (ctypes.c_int64 * len(PROF_KINDS))()
  check(lib().hsgk_profile_collect(ms, cnt))
  return {k: (ms[i], cnt[i]) for i, k in enumerate(PROF_KINDS)}


def verify_enable(on):
  lib().hsgk_verify_enable(int(bool(on)))


def verify_collect():
  """(rows compared, rows whose filtered label differs from the exact E-step) since the last
  collect, on the current device; synchronises it."""
  a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
  check(lib().hsgk_verify_collect(ctypes.byref(a), ctypes.byref(b)))
  return int(a.value), int(b.value)


# ---- device-side error flags ------------------------------------------------
# Data-dependent errors (a label outside [0, P)) are detected by the kernels.
# * The reference-shaped functional entry points (calculate_prototypes_from_labels with an explicit max_label)
#   read the flag AT THE CALL and raise there, as the reference does on CPU inside scatter_add_ (one host sync per
#   call; HSGK_SYNC_ERRORS=0 opts out).
# * The model-internal callers (labels in range by construction) and everything under HSGK_SYNC_ERRORS=0 keep the
#   host out of it: the flag is copied to pinned host memory behind the kernels and looked at by later libhsgk calls
#   once its event has completed -- the error surfaces asynchronously, like the reference's device-side assert on
#   a GPU.  HSGK_SYNC_ERRORS=1 checks immediately everywhere.
_deferred_lock = threading.Lock()
_deferred = []          # (event, pinned int32[1], message)


def sync_errors_default():
  """Whether the reference-shaped functional entry points read their error flag AT the call (the reference on CPU
  raises inside `scatter_add_`): yes unless HSGK_SYNC_ERRORS=0."""
  return os.environ.get('HSGK_SYNC_ERRORS') != '0'


def defer_status(status_dev, message, at_call=False):
  """at_call: read the flag now (one host sync) unless HSGK_SYNC_ERRORS=0; otherwise the flag travels behind the
  kernels and a later libhsgk call raises (at once with HSGK_SYNC_ERRORS=1)."""
  import torch
  env = os.environ.get('HSGK_SYNC_ERRORS')
  if env == '1' or (at_call and env != '0'):
    if int(status_dev.item()) != 0:
      raise HsgkError(message)
    return
  host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
  host.copy_(status_dev, non_blocking=True)
  ev = torch.cuda.Event()
  ev.record()
  with _deferred_lock:
    _deferred.append((ev, host, message))


def poll_deferred(wait=False):
  """Raises the recorded device-side errors whose kernels have finished (all of them with
  wait=True, which synchronises) as ONE HsgkError carrying every message."""
  with _deferred_lock:
    pending, done = [], []
    for item in _deferred:
      if wait:
        item[0].synchronize()
      (done if item[0].query() else pending).append(item)
    _deferred[:] = pending
  failed = [message for _ev, host, message in done if int(host[0]) != 0]
  if failed:
    raise HsgkError('; '.join(dict.fromkeys(failed)))


def _flush_deferred_at_exit():
  """An error flagged by the last calls of a run still surfaces (as a message: raising here would be lost)."""
  try:
    poll_deferred(wait=True)
  except HsgkError as e:
    import sys
    sys.stderr.write('hsg_amd: device-side error reported at exit: %s\n' % e)
  except Exception:                       # noqa: BLE001  (interpreter shutting down: the runtime may be gone)
    pass


import atexit   # noqa: E402
atexit.register(_flush_deferred_at_exit)


def stream_ptr():
  import torch
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
