"""The clustering method of the reference's stage-1 / inference embedding model
(hsg/models/embeddings/resnet_fcn.py: backbone + `segment_by_kmeans` only): `generate_clusters`
(:90-148) with the reference's arguments and output dict, bound onto the reference class by
`hsg_amd.patch_reference()` or mixed in:

    class ResnetFcn(SegsortClusteringMixin, reference.ResnetFcn): pass
"""
from hsg_amd.models.embeddings.resnet_fcn_hsg import _labels_and_ignore
from hsg_amd.utils.segsort import common as segsort_common


def generate_clusters(self, embeddings, semantic_labels, instance_labels, local_features=None):
  """`resnet_fcn.ResnetFcn.generate_clusters` (:90-148): the combined label map with its ignore value, spherical
  k-means of every image on libhsgk, the cluster labels split back into semantic / instance parts."""
  labels, ignore_index = _labels_and_ignore(self, semantic_labels, instance_labels)
  (cluster_embeddings, cluster_embeddings_with_loc, cluster_labels, cluster_indices,
   cluster_batch_indices) = segsort_common.segment_by_kmeans(
       embeddings, labels, self.kmeans_num_clusters, local_features=local_features,
       ignore_index=ignore_index, iterations=self.kmeans_iterations)
  return {
      'cluster_embedding': cluster_embeddings,
      'cluster_embedding_with_loc': cluster_embeddings_with_loc,
      'cluster_semantic_label': cluster_labels // self.label_divisor,
      'cluster_instance_label': cluster_labels % self.label_divisor,
      'cluster_index': cluster_indices,
      'cluster_batch_index': cluster_batch_indices,
  }


class SegsortClusteringMixin:
  """Method of `resnet_fcn.ResnetFcn` that sits on the hot path (mix in before the reference class)."""
  generate_clusters = generate_clusters
