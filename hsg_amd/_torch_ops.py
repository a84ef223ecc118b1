"""The torch-extension binding of libhsgk (hsg_amd/csrc/torch_ops.cpp -> libhsgk_torch.so): `TORCH_LIBRARY(hsgk, ...)`
ops whose backward is a C++ autograd node.  One dispatch per call where the ctypes mirror issues the
library call plus a handful of ATen ops from Python; both bindings drive the SAME kernels of libhsgk.so -- neither
is a fallback for missing device code.

`HSGK_BINDING=ctypes` keeps every call on the ctypes mirror (A/B, and the route the world > 1 exchange always
takes: its collectives sit between the phases)."""
import os
import threading

_lock = threading.Lock()
_state = {'tried': False, 'ops': None}
SO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'libhsgk_torch.so')


def ops():
  """torch.ops.hsgk once libhsgk_torch.so is loaded, else None (not built, or HSGK_BINDING=ctypes).

  Every call site fetches the namespace through here right before it dispatches an op, so this is also where the
  torch-bound route drains the deferred device flags (`_lib.check` does it for the ctypes route): a small-map
  time-out raises from the next libhsgk call of either binding, and the pinned words do not pile up."""
  if os.environ.get('HSGK_BINDING') == 'ctypes':
    return None
  if _state['tried']:
    from hsg_amd import _lib
    if _lib._deferred:
      _lib.poll_deferred()
  if not _state['tried']:
    with _lock:
      if not _state['tried']:
        import torch
        from hsg_amd import _lib
        _lib.lib()                                   # libhsgk.so first (same instance for both bindings)
        if os.path.exists(SO_PATH):
          torch.ops.load_library(SO_PATH)
          if int(torch.ops.hsgk.abi_version()) != _lib.ABI_VERSION:
            raise _lib.HsgkError('%s was built against another libhsgk (rebuild: make -C hsg_amd/csrc torch)' % SO_PATH)
          _state['ops'] = torch.ops.hsgk
        _state['tried'] = True
  return _state['ops']
