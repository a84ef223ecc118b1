"""The clustering half of the reference's embedding models
(hsg/models/embeddings/resnet_fcn_hsg.py) on libhsgk: `generate_clusters` and the methods
it calls, with the reference's names, arguments, result tuples and dict keys, for both
`ResnetFcn` (:130-301, :455-780) and `MultiviewResnetFcn` (:786-968, :1005-1136 -- the
model `pyscripts/train/train.py` builds).

The backbone, the position / query embeddings and the transformer stacks are the reference
model's own modules (stock PyTorch-ROCm, out of scope); the functions below only read them
from `self` (`fine_hrchy_transformer`, `fine_query_embed`, `label_divisor`, ...).  They are
bound onto the reference classes by `hsg_amd.patch_reference()`, or mixed in:

    class MultiviewResnetFcn(MultiviewClusteringMixin, reference.MultiviewResnetFcn): pass
"""
import torch

from hsg_amd.models.embeddings import hierarchy
from hsg_amd.utils.segsort import common as segsort_common


_IGNORE_SENTINEL = (1 << 62) - 1


def _labels_and_ignore(self, semantic_labels, instance_labels):
  """:189-197 / :853-861: combined label map and the value that marks ignored pixels."""
  if semantic_labels is None or instance_labels is None:
    return None, None
  labels = semantic_labels * self.label_divisor + instance_labels
  # The reference marks ignored pixels with `labels.max() + 1`, a device scalar that segment_by_kmeans would have
  # to read back (a stalling read per step).  The marked pixels are dropped before anything else looks at their
  # value, so any value no kept pixel carries gives the same outputs: a host constant beyond every label.
  ignore_index = _IGNORE_SENTINEL
  labels = labels.masked_fill(semantic_labels == self.semantic_ignore_index, ignore_index)
  return labels, ignore_index


def _calculate_kmeans_prototypes(self, cluster_embeddings, cluster_indices, cluster_batch_indices,
                                 cluster_pos_embeddings, cluster_labels, image_indices=None):
  """:455-577 (no image_indices) and :1005-1136 (multiview)."""
  return hierarchy.calculate_kmeans_prototypes(
      cluster_embeddings, cluster_indices, cluster_batch_indices, cluster_pos_embeddings,
      cluster_labels, image_indices, label_divisor=self.label_divisor,
      max_num_clusters=None if getattr(self, 'dynamic_max_num_clusters', False) else self.max_num_clusters)


def _collect_nd_coarser_prototype(self, prototypes, prototype_grouping_labels,
                                  prototype_padding_masks=None, num_groups=None, normalized=True):
  """:683-748."""
  return hierarchy.collect_nd_coarser_prototype(prototypes, prototype_grouping_labels,
                                                prototype_padding_masks, num_groups, normalized)


def _collect_pixel_hierarchical_clustering_indices(self, cluster_indices_by_batch,
                                                   cluster_batch_indices,
                                                   finehrchy_prototype_grouping_labels):
  """:751-780."""
  return hierarchy.collect_pixel_hierarchical_clustering_indices(
      cluster_indices_by_batch, cluster_batch_indices, finehrchy_prototype_grouping_labels)


def _hierarchical_grouping(self, prototypes, pos_prototypes, prototype_padding_masks):
  """:580-681: two clustering transformers (the model's own modules) and, on libhsgk, the
  softmax / argmax / Bayes-chained coarse assignment (`hsgk_hier_assign`) and the masked
  group means between them (`hsgk_group_mean`).  Same 8-tuple as the reference."""
  fine_query_embed = self.fine_query_embed()
  (fine_centroids, fine_centroid_feats, fine_logits, fine_memory) = self.fine_hrchy_transformer(
      src=prototypes, mask=prototype_padding_masks, query_embed=fine_query_embed,
      pos_embed=pos_prototypes)
  fine_labels, fine_probs, _, _ = hierarchy.hierarchical_grouping_from_logits(fine_logits)      # :638-641
  fine_pos_prototypes = _collect_nd_coarser_prototype(                                          # :644-649
      self, pos_prototypes, fine_labels, prototype_padding_masks,
      num_groups=self.fine_hrchy_clusters, normalized=False)
  coarse_query_embed = self.coarse_query_embed()
  (coarse_centroids, _coarse_feats, coarse_logits, coarse_memory) = self.coarse_hrchy_transformer(
      src=fine_centroid_feats, mask=None, query_embed=coarse_query_embed,
      pos_embed=fine_pos_prototypes)
  # :662-672: softmax over the coarse clusters, chained with the fine probabilities, argmax
  _, fine_probs, coarse_labels, coarse_probs = hierarchy.hierarchical_grouping_from_logits(
      fine_logits, coarse_logits)
  return (fine_labels, fine_centroids, fine_probs, fine_memory,
          coarse_labels, coarse_centroids, coarse_probs, coarse_memory)


def _generate_clusters(self, embeddings, semantic_labels, instance_labels, image_indices,
                       local_features, pos_embeddings, multiview):
  labels, ignore_index = _labels_and_ignore(self, semantic_labels, instance_labels)
  # Step 1 (:199-212): spherical k-means aligned with the label map
  (cluster_embeddings, cluster_embeddings_with_loc, cluster_labels, cluster_indices,
   cluster_batch_indices) = segsort_common.segment_by_kmeans(
       embeddings, labels, self.kmeans_num_clusters, local_features=local_features,
       ignore_index=ignore_index, iterations=self.kmeans_iterations)
  cluster_semantic_labels = cluster_labels // self.label_divisor
  cluster_instance_labels = cluster_labels % self.label_divisor

  # :217-226: position embeddings of the kept pixels (they are not normalised, so they do
  # not travel through segment_by_kmeans)
  if pos_embeddings is not None and labels is not None:
    # (the number of kept pixels is the length of the k-means outputs: no host read for the size of `nonzero`)
    valid_pixels = torch.nonzero_static((labels != ignore_index).view(-1),
                                        size=int(cluster_embeddings.shape[0])).view(-1)
    flat_pos = pos_embeddings.permute(0, 2, 3, 1).contiguous().flatten(0, 2)
    cluster_pos_embeddings = torch.index_select(flat_pos, 0, valid_pixels)
  else:
    cluster_pos_embeddings = None

  # Step 2 (:228-242): padded per-image segment prototypes
  if multiview:
    protos = self._calculate_kmeans_prototypes(
        cluster_embeddings, cluster_indices, cluster_batch_indices, cluster_pos_embeddings,
        cluster_labels, image_indices)
  else:
    protos = self._calculate_kmeans_prototypes(
        cluster_embeddings, cluster_indices, cluster_batch_indices, cluster_pos_embeddings,
        cluster_labels)
  (prototypes, pos_prototypes, prototype_padding_masks, prototype_labels, prototype_batch_indices,
   cluster_indices_by_image) = protos
  prototype_semantic_labels = prototype_labels // self.label_divisor
  prototype_instance_labels = prototype_labels % self.label_divisor

  # Step 3 (:244-255): hierarchical grouping with the clustering transformers
  (fine_labels, fine_centroids, fine_logits, fine_memory, coarse_labels, coarse_centroids,
   coarse_logits, coarse_memory) = self._hierarchical_grouping(
       prototypes, pos_prototypes, prototype_padding_masks)

  # :257-267 / :942-957: pixel-wise fine / coarse ids (keyed by image id in the multiview model)
  if multiview and image_indices is not None:
    pixel_image_indices = torch.gather(image_indices, 0, cluster_batch_indices)
  else:
    pixel_image_indices = cluster_batch_indices
  # (the dense image number of every pixel and the image-by-image order were found while the prototype tables
  #  were built; they ride on `cluster_indices_by_image` -- ops.note -- instead of a second `unique` + order check.
  #  This caller vouches that `pixel_image_indices` is the id vector those tables were built from; any other
  #  caller of the method gets the reference's lookup from the ids it passes)
  hierarchy.vouch_pixel_images(cluster_indices_by_image, pixel_image_indices)
  finehrchy_cluster_indices = self._collect_pixel_hierarchical_clustering_indices(
      cluster_indices_by_image, pixel_image_indices, fine_labels)
  coarsehrchy_cluster_indices = self._collect_pixel_hierarchical_clustering_indices(
      cluster_indices_by_image, pixel_image_indices, coarse_labels)

  return {
      'cluster_embedding': cluster_embeddings,
      'cluster_embedding_with_loc': cluster_embeddings_with_loc,
      'cluster_semantic_label': cluster_semantic_labels,
      'cluster_instance_label': cluster_instance_labels,
      'cluster_index': cluster_indices,
      'cluster_batch_index': cluster_batch_indices,
      'finehrchy_cluster_index': finehrchy_cluster_indices,
      'coarsehrchy_cluster_index': coarsehrchy_cluster_indices,
      'nd_prototype': prototypes,
      'nd_prototype_padding_mask': prototype_padding_masks,
      'nd_prototype_batch_index': prototype_batch_indices,
      'nd_prototype_semantic_label': prototype_semantic_labels,
      'nd_prototype_instance_label': prototype_instance_labels,
      'cluster_index_by_image': cluster_indices_by_image,
      'finehrchy_nd_prototype_grouping_label': fine_labels,
      'finehrchy_nd_prototype_grouping_centroid': fine_centroids,
      'finehrchy_nd_prototype_grouping_logit': fine_logits,
      'finehrchy_nd_prototype_encoder_memory': fine_memory,
      'coarsehrchy_nd_prototype_grouping_label': coarse_labels,
      'coarsehrchy_nd_prototype_grouping_centroid': coarse_centroids,
      'coarsehrchy_nd_prototype_grouping_logit': coarse_logits,
      'coarsehrchy_nd_prototype_encoder_memory': coarse_memory,
  }


def generate_clusters(self, embeddings, semantic_labels, instance_labels, local_features=None,
                      pos_embeddings=None):
  """`ResnetFcn.generate_clusters` (:130-301): same arguments and output dict."""
  return _generate_clusters(self, embeddings, semantic_labels, instance_labels, None,
                            local_features, pos_embeddings, multiview=False)


def generate_clusters_multiview(self, embeddings, semantic_labels, instance_labels, image_indices,
                                local_features=None, pos_embeddings=None):
  """`MultiviewResnetFcn.generate_clusters` (:786-968): same arguments and output dict."""
  return _generate_clusters(self, embeddings, semantic_labels, instance_labels, image_indices,
                            local_features, pos_embeddings, multiview=True)


class ClusteringMixin:
  """Methods of `ResnetFcn` that sit on the hot path (mix in before the reference class)."""
  generate_clusters = generate_clusters
  _calculate_kmeans_prototypes = _calculate_kmeans_prototypes
  _hierarchical_grouping = _hierarchical_grouping
  _collect_nd_coarser_prototype = _collect_nd_coarser_prototype
  _collect_pixel_hierarchical_clustering_indices = _collect_pixel_hierarchical_clustering_indices


class MultiviewClusteringMixin(ClusteringMixin):
  """Methods of `MultiviewResnetFcn` (the model train.py builds)."""
  generate_clusters = generate_clusters_multiview


class ClusteringMixinCs(ClusteringMixin):
  """`resnet_fcn_hsg_cs.ResnetFcn` (the Cityscapes twin): the same methods, the padded
  tables as long as the largest number of clusters of an image in the call (:499-502) instead of
  `self.max_num_clusters`."""
  dynamic_max_num_clusters = True


class MultiviewClusteringMixinCs(MultiviewClusteringMixin):
  """`resnet_fcn_hsg_cs.MultiviewResnetFcn` (:1061-1064)."""
  dynamic_max_num_clusters = True
