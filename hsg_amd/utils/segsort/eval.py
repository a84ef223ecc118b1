"""Drop-in for `hsg.utils.segsort.eval` (reference hsg/utils/segsort/eval.py).

`top_k_ranking` is evaluated every training step for the accuracy metric
(hsg/models/predictions/hsg.py:113-118) and is the retrieval step of inference
(hsg/models/predictions/segsort.py:66-123).  The reference materialises the
[N,P] affinity matrix and argsorts each row; libhsgk streams the queries against
64-prototype blocks on fp32 MFMA and keeps only the k best per row.
"""
import torch

from hsg_amd import _lib, _torch_ops, ops
from hsg_amd.utils.general import common as common_utils


def top_k_indices(embeddings, prototypes, top_k, query_groups=None, prototype_groups=None):
  """[N, top_k] prototype indices by descending inner product (and the products).  With the
  two group vectors only prototypes of the query's own group compete; unfilled slots come
  back as index 0 with value -inf."""
  ops.require_gpu(embeddings, 'embeddings')
  tops = _torch_ops.ops()
  if tops is not None:                 # the torch-extension binding: one dispatch
    return tops.topk_prototypes(embeddings, prototypes, int(top_k), query_groups, prototype_groups)
  q = embeddings.detach().reshape(-1, embeddings.shape[-1]).float().contiguous()
  p = prototypes.detach().reshape(-1, prototypes.shape[-1]).float().contiguous()
  n, c = q.shape
  L = _lib.lib()
  with torch.cuda.device(q.device):
    idx = torch.empty((n, top_k), dtype=torch.long, device=q.device)
    val = torch.empty((n, top_k), dtype=torch.float32, device=q.device)
    wsb = L.hsgk_topk_workspace_bytes(n, c, p.shape[0], int(top_k))
    ws = torch.empty((wsb,), dtype=torch.uint8, device=q.device)
    qg = pg = None
    if query_groups is not None:
      qg = query_groups.reshape(-1).to(torch.int64).contiguous()
      pg = prototype_groups.reshape(-1).to(torch.int64).contiguous()
    _lib.check(L.hsgk_topk_prototypes_grouped(
        q.data_ptr(), n, c, p.data_ptr(), p.shape[0], int(top_k),
        qg.data_ptr() if qg is not None else None, pg.data_ptr() if pg is not None else None,
        idx.data_ptr(), val.data_ptr(), ws.data_ptr(), wsb, _lib.stream_ptr()))
  return idx, val


def top_k_ranking(embeddings, labels, prototypes, prototype_labels, top_k=3):
  """Top-k retrieval accuracy and the retrieved labels (reference eval.py:9-52)."""
  idx, _ = top_k_indices(embeddings, prototypes, top_k)
  plab = prototype_labels.view(-1)
  top_k_labels = plab[idx.view(-1)].view(-1, top_k)
  hits = torch.eq(labels.view(-1, 1), top_k_labels)
  return torch.mean(hits.float()), top_k_labels


def majority_label_from_topk(top_k_labels, num_classes=None):
  """Most frequent retrieved label per query (reference eval.py:55-70)."""
  counts = torch.sum(common_utils.one_hot(top_k_labels, num_classes), dim=1)
  return torch.argmax(counts, 1)
