"""`TransformerClustering.forward` of the reference
(hsg/models/embeddings/transformer_clusters.py:60-114) with its tail -- logits, the
`num_clusters` queries of largest activation, the three gathers -- on libhsgk
(`hsgk_cluster_topk`).  The transformer stack and the two FC + BN heads are the module's own
(stock PyTorch-ROCm); `forward` below is bound onto the reference class by
`hsg_amd.patch_reference()` or inherited through `TransformerClusteringMixin`.
"""
from hsg_amd.models.embeddings import hierarchy


def forward(self, src, mask, query_embed, pos_embed):
  """Same arguments and results as the reference: src [B,C,sl], mask [B,sl] bool, query_embed
  [tl,C] or [B,C,tl], pos_embed [B,C,sl] -> (centroids [B,C,k], centroid_feats [B,C,k],
  logits [B,k,sl], node_features [B,C,sl])."""
  bs, cs, _ = src.shape
  centroids, node_features = self._transformer(src, mask, query_embed, pos_embed)   # :81-84
  tl = centroids.shape[-1]
  flat_centroids = centroids.transpose(1, 2).flatten(0, 1)                          # :86
  centroids = self.centroid_fc(flat_centroids).view(bs, tl, cs).transpose(1, 2)
  centroid_feats = self.centroid_feat_fc(flat_centroids).view(bs, tl, cs).transpose(1, 2)
  c_sel, cf_sel, logits, _order = hierarchy.transformer_clustering_tail(               # :91-112
      centroids, centroid_feats, node_features, self._num_clusters)
  return c_sel, cf_sel, logits, node_features


class TransformerClusteringMixin:
  """class TransformerClustering(TransformerClusteringMixin, reference.TransformerClustering)"""
  forward = forward
