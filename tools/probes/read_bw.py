"""Achievable HBM read bandwidth on this box: torch reductions over 4.93 GB (the fp16 copy's size) in a few dtypes."""
import torch, time
dev = torch.device('cuda:0')
n = 4_930_000_000
for dt, name in ((torch.float32, 'f32 sum'), (torch.float16, 'f16 sum'), (torch.int32, 'i32 max')):
  x = torch.empty(n // torch.empty((), dtype=dt).element_size(), dtype=dt, device=dev)
  x.zero_()
  f = (lambda: x.max()) if dt == torch.int32 else (lambda: x.sum())
  for _ in range(2): f()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(10): f()
  b.record(); torch.cuda.synchronize()
  ms = a.elapsed_time(b) / 10
  print('%-8s %.3f ms  %.2f TB/s' % (name, ms, n / ms / 1e9))
  del x
y = torch.empty(n // 4, dtype=torch.float32, device=dev); z = torch.empty_like(y)
for _ in range(2): z.copy_(y)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): z.copy_(y)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print('copy     %.3f ms  %.2f TB/s (read + write)' % (ms, 2 * n / ms / 1e9))
w = torch.empty(n // 4, dtype=torch.float32, device=dev)
for _ in range(2): w.fill_(1.0)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): w.fill_(1.0)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print('fill     %.3f ms  %.2f TB/s (write only)' % (ms, n / ms / 1e9))
