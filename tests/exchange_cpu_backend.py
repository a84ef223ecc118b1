"""Oracle-backed stand-in for hsg_amd.models.utils.HsgkExchangeBackend (TEST infrastructure): the same
phases -- keys / merge / sums / finish and their backward pieces -- restated with numpy + the CPU oracle's
segment sums, with the block layout of hsg_amd/csrc/exchange.hip (8 header words: [0] row count, [1] error
bits; then the sorted (batch, cluster, semantic, instance) tuples).  Installed by the gloo tests so that the
ORCHESTRATION of hsg_amd/models/utils.py (capacity protocol, the two collectives, the list mode of one
process driving several devices) runs on CPU exactly as it does over RCCL."""
import ctypes

import numpy as np
import torch

from oracle import oracle as orc

HDR = 8


def _segment_sums(x, ids, count):
  x = np.ascontiguousarray(x, np.float32)
  lab = np.ascontiguousarray(ids, np.int64)
  out = np.zeros((count, x.shape[1]), np.float32)
  if count and x.shape[0]:
    orc.lib().orc_segment_sums(
        x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.c_int64(x.shape[0]), x.shape[1],
        lab.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int64(count), orc.CHUNK,
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
  return out


class CpuExchangeBackend:
  def __init__(self, emb, emb_loc, c, b, sem, inst, cap_local, cap_total, world):
    self.emb = emb.detach().reshape(-1, emb.shape[-1]).numpy()
    self.emb_loc = emb_loc.detach().reshape(-1, emb_loc.shape[-1]).numpy()
    self.tuples_in = np.stack([t.detach().reshape(-1).numpy().astype(np.int64) for t in (b, c, sem, inst)], 1)
    self.n, self.C, self.D = self.emb.shape[0], self.emb.shape[1], self.emb_loc.shape[1]
    self.cap, self.cap_total, self.world = int(cap_local), int(cap_total), int(world)
    self.table = torch.zeros((self.cap_total, self.C + self.D), dtype=torch.float32)
    self.upd = torch.zeros((self.n,), dtype=torch.int64)
    self.plab = torch.zeros((3, self.cap_total), dtype=torch.int64)
    self.meta = [0, 0, 0, 0]
    self._recv = torch.zeros((self.world, HDR + 4 * self.cap), dtype=torch.int64)
    self._send = torch.zeros((HDR + 4 * self.cap,), dtype=torch.int64)
    self._slots = None

  def keys(self):
    err = 1 if (self.n and self.tuples_in.min() < 0) else 0
    if self.n:
      uniq, inv = np.unique(self.tuples_in, axis=0, return_inverse=True)      # lexicographic rows
    else:
      uniq, inv = np.zeros((0, 4), np.int64), np.zeros((0,), np.int64)
    self.local_ids = inv.reshape(-1)
    cnt = uniq.shape[0]
    if cnt > self.cap:
      err |= 2
    m = min(cnt, self.cap)
    self._send.zero_()
    self._send[0], self._send[1] = cnt, err
    self._send[HDR:HDR + 4 * m] = torch.from_numpy(uniq[:m].reshape(-1))
    if self.world == 1:
      self._recv[0].copy_(self._send)
    return self._send

  def recv_blocks(self):
    return self._recv

  def merge(self, my_rank):
    recv = self._recv.numpy()
    err, lists = 0, []
    for r in range(self.world):
      cnt = int(recv[r, 0])
      err |= int(recv[r, 1])
      if cnt > self.cap:
        err |= 2
      m = min(cnt, self.cap)
      lists.append(recv[r, HDR:HDR + 4 * m].reshape(m, 4))
    allt = np.concatenate(lists, 0) if lists else np.zeros((0, 4), np.int64)
    glob = np.unique(allt, axis=0) if allt.shape[0] else allt
    total = glob.shape[0]
    if total > self.cap_total:
      err |= 8
    slots = torch.zeros((self.world, self.cap), dtype=torch.int32)
    index = {tuple(t): i for i, t in enumerate(glob.tolist())}
    for r, lst in enumerate(lists):
      for i, t in enumerate(lst.tolist()):
        slots[r, i] = min(index[tuple(t)], self.cap_total - 1)
    keep = min(total, self.cap_total)
    self.plab[2, :keep] = torch.from_numpy(glob[:keep, 0])       # batch
    self.plab[0, :keep] = torch.from_numpy(glob[:keep, 2])       # semantic
    self.plab[1, :keep] = torch.from_numpy(glob[:keep, 3])       # instance
    self.meta = [int(recv[my_rank, 0]) if my_rank >= 0 else 0, total, err,
                 max(int(recv[r, 0]) for r in range(self.world))]
    self._slots = slots
    return slots

  def sums(self, my_rank, slots_row=None):
    if slots_row is None:
      slots_row = self._slots[max(my_rank, 0)]
    if self.meta[2] and self._slots is not None:
      return
    row = slots_row.numpy().astype(np.int64)
    upd = row[np.minimum(self.local_ids, max(len(row) - 1, 0))] if self.n else np.zeros((0,), np.int64)
    self.upd = torch.from_numpy(upd)
    tab = np.concatenate([_segment_sums(self.emb, upd, self.cap_total),
                          _segment_sums(self.emb_loc, upd, self.cap_total)], 1)
    self.table = torch.from_numpy(tab)

  def read_meta(self):
    return tuple(self.meta)

  def finish(self, table_rows):
    C = self.C
    na = table_rows[:, :C].norm(dim=1, keepdim=True).clamp_min(1e-12)
    nb = table_rows[:, C:].norm(dim=1, keepdim=True).clamp_min(1e-12)
    return table_rows[:, :C] / na, table_rows[:, C:] / nb, torch.cat([na, nb], 1)

  @staticmethod
  def finish_bwd(g_pa, g_pb, pa, pb, norms):
    parts = []
    for g, out, col in ((g_pa, pa, 0), (g_pb, pb, 1)):
      if g is None:
        parts.append(torch.zeros_like(out))
        continue
      dot = (g * out).sum(1, keepdim=True)
      parts.append((g - out * dot) / norms[:, col:col + 1])
    return torch.cat(parts, 1)

  @staticmethod
  def rows_bwd(g_table, upd, C, D, need):
    return [g_table[:, :C][upd] if need[0] else None, g_table[:, C:][upd] if need[1] else None]
