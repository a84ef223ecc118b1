"""Portable synthetic pixel-embedding generator.

Every value is produced with integer arithmetic only (a splitmix64 hash of the
flat element index) followed by one exact power-of-two scaling, so numpy on any
host reproduces the same float32 bit patterns.  Golden fixtures therefore
store only *outputs*; inputs are regenerated from (seed, shape, flavour).

Flavours:
  'iid'      sum of four 16-bit uniforms (Irwin-Hall, ~N(0,1) after scaling)
  'mixture'  per image: 20 unit centres + 0.05 * iid noise, picked per pixel
             in spatially coherent blobs (k-means converges, clusters empty out)
"""
import numpy as np

SEED_BASE = 0x48534700  # 'HSG\0'; BASELINE.md: seed = SEED_BASE + cfg id

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
  """Vectorised splitmix64 finaliser on uint64 arrays."""
  x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
  z = x
  z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
  z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
  return z ^ (z >> np.uint64(31))


def hash_u64(seed, n, offset=0):
  """n hashed uint64 words for element indices offset..offset+n-1."""
  with np.errstate(over='ignore'):
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    key = _splitmix64(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(1))
    return _splitmix64(idx ^ key)


def gaussish(seed, n, offset=0):
  """Approximately N(0,1) float32 values, exactly reproducible.

  Sum of the four 16-bit fields of one hash word, centred, times 2**-16 * sqrt(3)
  rounded to the float32 constant below (one exact-product rounding only, which
  IEEE-754 makes identical everywhere).
  """
  h = hash_u64(seed, n, offset)
  s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
       + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48)))
  centred = s.astype(np.int64) - np.int64(2 * 65535)          # exact integer
  # var of one 16-bit uniform = (2^32-1)/12; four of them -> std ~ 37837.2
  return (centred.astype(np.float32) * np.float32(1.0 / 37837.2)).astype(np.float32)


def embeddings_nchw(seed, shape, flavour='iid'):
  """Synthetic [B,C,H,W] float32 embeddings."""
  B, C, H, W = shape
  n = B * C * H * W
  if flavour == 'iid':
    return gaussish(seed, n).reshape(B, C, H, W)
  if flavour != 'mixture':
    raise ValueError('flavour must be iid or mixture')
  ncentres = 20
  cen = mixture_centres(seed, B, C, ncentres)
  # blob map: centre id depends on a coarse 2-D cell hashed per image
  cell = 8
  gy = (np.arange(H) // max(1, H // cell)).astype(np.uint64)
  gx = (np.arange(W) // max(1, W // cell)).astype(np.uint64)
  out = np.empty((B, C, H, W), np.float32)
  noise = gaussish(seed ^ 0xA5A5, n).reshape(B, C, H, W)
  for b in range(B):
    cid = _splitmix64((gy[:, None] * np.uint64(131) + gx[None, :]) * np.uint64(2654435761)
                      + np.uint64(b * 7919 + seed)) % np.uint64(ncentres)
    picked = cen[b][cid.astype(np.int64)]                    # [H,W,C]
    out[b] = picked.transpose(2, 0, 1) + np.float32(0.05) * noise[b]
  return out


def overseg_labels(seed, B, H, W, regions=48, ignore_rows=4, ignore_index=255):
  """Random blocky over-segmentation [B,H,W] int64 with an ignore band.

  Region ids are < regions and never equal ignore_index unless regions > ignore_index.
  """
  lab = np.empty((B, H, W), np.int64)
  ry = max(1, H // 8)
  rx = max(1, W // 8)
  yy = (np.arange(H) // ry).astype(np.uint64)
  xx = (np.arange(W) // rx).astype(np.uint64)
  for b in range(B):
    h = _splitmix64((yy[:, None] * np.uint64(977) + xx[None, :]) + np.uint64(seed + 31 * b))
    lab[b] = (h % np.uint64(regions)).astype(np.int64)
    if ignore_rows:
      lab[b, :ignore_rows, :] = ignore_index
  return lab


def stream_key(seed):
  """The 64-bit key hash_u64 mixes into every element index of stream `seed`."""
  with np.errstate(over='ignore'):
    return int(_splitmix64(np.uint64(seed) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(1)))


def mixture_centres(seed, B, C, ncentres=20):
  """[B, ncentres, C] float32 unit centres of the 'mixture' flavour."""
  cen = gaussish(seed ^ 0x5A5A, B * ncentres * C).reshape(B, ncentres, C).astype(np.float64)
  cen /= np.sqrt((cen * cen).sum(-1, keepdims=True))
  return cen.astype(np.float32)


def device_embeddings_nchw(seed, shape, flavour='iid', device='cuda', first_image=0):
  """embeddings_nchw(seed, (first_image + B, C, H, W), flavour)[first_image:] generated in
  HBM by libhsgk (hsgk_synth_gaussish / hsgk_synth_mixture): bit-identical to the numpy
  generator, so a full-size BASELINE batch needs no host memory and no PCIe transfer.
  `first_image` is the global index of this rank's first image (per-GPU shard
  offset = global image index)."""
  import ctypes
  import torch
  from hsg_amd import _lib
  B, C, H, W = shape
  dev = torch.device(device)
  L = _lib.lib()
  with torch.cuda.device(dev):
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=dev)
    offset = first_image * C * H * W
    if flavour == 'iid':
      _lib.check(L.hsgk_synth_gaussish(ctypes.c_uint64(stream_key(seed)), ctypes.c_uint64(offset),
                                       out.numel(), out.data_ptr(), _lib.stream_ptr()))
      return out
    if flavour != 'mixture':
      raise ValueError('flavour must be iid or mixture')
    cen = torch.from_numpy(mixture_centres(seed, first_image + B, C)[first_image:].copy()).to(dev)
    _lib.check(L.hsgk_synth_mixture(ctypes.c_uint64(stream_key(seed ^ 0xA5A5)), ctypes.c_uint64(seed),
                                    cen.data_ptr(), cen.shape[1], first_image, B, C, H, W,
                                    out.data_ptr(), _lib.stream_ptr()))
    torch.cuda.current_stream().synchronize()      # `cen` dies here
  return out
