"""Probe (GPU box): can several RCCL ranks share ONE device?  And does gloo take device tensors?
Prints one line per finding; never raises."""
import os
import sys
import socket
import traceback

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def worker(rank, world, port, backend):
  import torch
  import torch.distributed as dist
  import datetime
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.cuda.set_device(0)
  dev = torch.device('cuda', 0)
  try:
    kw = {'device_id': dev} if backend == 'nccl' else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60), **kw)
    t = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    g = torch.empty((world * 4,), device=dev)
    dist.all_gather_into_tensor(g, torch.full((4,), float(rank), device=dev))
    torch.cuda.synchronize()
    if rank == 0:
      print('PROBE %s world=%d on one device: all_reduce -> %s, all_gather -> %s' % (backend, world, t[0].item(), g.tolist()), flush=True)
    dist.destroy_process_group()
  except Exception as e:      # noqa: BLE001
    print('PROBE %s world=%d rank %d FAILED: %s: %s' % (backend, world, rank, type(e).__name__, str(e)[:300]), flush=True)


if __name__ == '__main__':
  import torch.multiprocessing as mp
  for backend in sys.argv[1:] or ['nccl', 'gloo']:
    for world in (2, 4):
      try:
        mp.spawn(worker, args=(world, free_port(), backend), nprocs=world, join=True)
      except Exception:       # noqa: BLE001
        print('PROBE %s world=%d spawn failed' % (backend, world))
        traceback.print_exc()
