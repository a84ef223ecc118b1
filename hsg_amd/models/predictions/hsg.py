"""Drop-in for `hsg.models.predictions.hsg.Hsg` (reference
hsg/models/predictions/hsg.py): same constructor (an easydict-style `config`), same
`losses` / `forward` results.

The three pixel-to-segment contrastive losses (image similarity :83-110, fine :120-138 and
coarse :140-161 hierarchy) contrast the SAME embeddings with the SAME prototype table and
differ only in their semantic labels; the reference evaluates E P^T three times, here they
are ONE libhsgk pass over three label sets (`segsort_losses`, csrc/loss.hip) whenever more
than one of them is enabled.  The clustering regularisers (DMon :163-186, centroid contrast
:188-227) use the hsg_amd versions of the same modules.
"""
import torch
import torch.nn as nn

import hsg_amd.utils.general.common as common_utils
import hsg_amd.utils.graph.loss as graph_loss
import hsg_amd.utils.segsort.eval as segsort_eval
import hsg_amd.utils.segsort.loss as segsort_loss


def _construct_loss(self, loss_types, **kwargs):
  """:59-71."""
  if loss_types == 'segsort':
    return segsort_loss.SegSortLoss(kwargs['concentration'], group_mode='segsort+',
                                    reduction='mean')
  elif loss_types == 'dmon':
    return graph_loss.DMonLoss(adj_knn=kwargs['adj_knn'])
  elif loss_types == 'none':
    return None
  else:
    raise KeyError('Unsupported loss types: {:s}'.format(loss_types))


def _is_segsort(mod):
  """A plain SegSortLoss with mean reduction -- ours or the reference's (duck-typed so that
  a model built by the reference's constructor is served too)."""
  return (mod is not None and type(mod).__name__ == 'SegSortLoss'
          and getattr(mod, 'reduction', 'mean') == 'mean'
          and hasattr(mod, 'concentration') and hasattr(mod, 'group_mode'))


_HIERARCHY_LEVELS = ('coarse', 'fine')


def _dmon_terms(self, datas):
  """Spectral (DMon) clustering objective + collapse regulariser of both hierarchy levels (reference :163-186):
  every level's grouping logits are scored against the same k-NN graph inputs (node prototypes, padding
  masks, node -> image indices)."""
  # (the Cityscapes twin, predictions/hsg_cs.py:173-175, builds the k-NN graph over ALL nodes of an image row:
  #  no per-view segments)
  segments = datas['nd_prototype_batch_index'] if getattr(self, 'dmon_graph_per_view', True) else None
  graph = (datas['nd_prototype'], datas['nd_prototype_padding_mask'], segments)
  # the k-NN graph depends on the nodes only: built once for both levels (the reference rebuilds it per call)
  extra = {'adjacency': self.dmon_loss.adjacency(*graph)} if hasattr(self.dmon_loss, 'adjacency') else {}
  total = None
  for level in _HIERARCHY_LEVELS:
    cut, collapse = self.dmon_loss(datas[level + 'hrchy_nd_prototype_grouping_logit'], *graph, **extra)
    total = cut + collapse if total is None else total + cut + collapse
  return total


def _centroid_rows(centroids):
  """[images, channels, groups] -> unit rows [images * groups, channels]."""
  return common_utils.normalize_embedding(centroids.permute(0, 2, 1).reshape(-1, centroids.shape[1]))


def _centroid_contrast_terms(self, datas, targets):
  """Group centroids of this GPU's images against the batch-wide target centroids (reference :188-227): the
  label of a centroid is its row in the target table, and this GPU's images are a contiguous range of that
  table -- found ONCE for both levels (the reference, and the round-2 mirror, re-derived it per level with
  its own host read)."""
  image_indices = torch.gather(targets['image_index'], 0, datas['cluster_batch_index'])
  first = image_indices.min()                 # stays on the device: the block's length is the local table's (no host read)
  total = None
  for level in _HIERARCHY_LEVELS:
    target = targets[level + 'hrchy_nd_prototype_grouping_centroid']
    images, groups = target.shape[0], target.shape[2]
    target_labels = torch.arange(images * groups, dtype=torch.long, device=target.device)
    # the reference slices target_labels[first * groups : (last + 1) * groups] (:203-206), one label per local
    # centroid row: images first .. last are this GPU's rows of the table, i.e. as many as the local table has
    own_rows = datas[level + 'hrchy_nd_prototype_grouping_centroid'].shape[0] * groups
    own_labels = target_labels[:own_rows] + first * groups
    loss = self.centroid_cont_loss(_centroid_rows(datas[level + 'hrchy_nd_prototype_grouping_centroid']),
                                   own_labels, own_labels, _centroid_rows(target), target_labels)
    total = loss if total is None else total + loss
  return total


def losses(self, datas, targets={}):
  """:78-227: (img_sim_loss, hrchy_group_loss, clustering_loss, img_sim_acc)."""
  img_sim_loss = None
  hrchy_group_loss = None
  clustering_loss = None
  img_sim_acc = None

  # ---- label sets of the three contrastive losses (:83-103, :120-128, :140-147)
  pending = []          # (name, module, weight, pixel labels, prototype labels)
  if self.img_sim_loss is not None:
    batch_indices = datas['cluster_batch_index']
    image_indices = torch.gather(targets['image_index'], 0, batch_indices)
    instance_labels = datas['cluster_instance_label'] * self.label_divisor + image_indices
    prototype_image_indices = torch.gather(targets['image_index'], 0,
                                           targets['prototype_batch_index'])
    prototype_instance_labels = (targets['prototype_instance_label'] * self.label_divisor
                                 + prototype_image_indices)
    pending.append(('img_sim', self.img_sim_loss, self.img_sim_loss_weight, instance_labels,
                    prototype_instance_labels))
  if self.fine_hrchy_loss is not None:
    plabs = targets['finehrchy_mapping_index']
    pending.append(('fine', self.fine_hrchy_loss, self.fine_hrchy_loss_weight,
                    torch.gather(plabs, 0, datas['cluster_index']), plabs))
  if self.coarse_hrchy_loss is not None:
    plabs = targets['coarsehrchy_mapping_index']
    pending.append(('coarse', self.coarse_hrchy_loss, self.coarse_hrchy_loss_weight,
                    torch.gather(plabs, 0, datas['cluster_index']), plabs))

  values = {}
  if pending:
    embeddings = datas['cluster_embedding']
    cluster_indices = datas['cluster_index']
    prototypes = targets['prototype']
    if len(pending) > 1 and all(_is_segsort(p[1]) for p in pending):
      outs = segsort_loss.segsort_losses(
          embeddings, cluster_indices, prototypes,
          [(p[3], p[4], p[1].concentration, p[1].group_mode) for p in pending])
      for p, v in zip(pending, outs):
        values[p[0]] = v * p[2]
    else:
      for name, mod, weight, labs, plabs in pending:
        values[name] = mod(embeddings, labs, cluster_indices, prototypes, plabs) * weight
  if 'img_sim' in values:
    img_sim_loss = values['img_sim']
    p = pending[0]
    img_sim_acc, _ = segsort_eval.top_k_ranking(prototypes, p[4], prototypes, p[4], 5)   # :113-118
  if 'fine' in values:
    hrchy_group_loss = values['fine']
  if 'coarse' in values:
    hrchy_group_loss = values['coarse'] if hrchy_group_loss is None else hrchy_group_loss + values['coarse']

  terms = []
  if self.dmon_loss is not None:
    terms.append(_dmon_terms(self, datas) * self.dmon_loss_weight)
  if self.centroid_cont_loss is not None:
    terms.append(_centroid_contrast_terms(self, datas, targets) * self.centroid_cont_loss_weight)
  if terms:
    clustering_loss = terms[0] if len(terms) == 1 else terms[0] + terms[1]

  return img_sim_loss, hrchy_group_loss, clustering_loss, img_sim_acc


class Hsg(nn.Module):
  """Losses of HSG (reference class `Hsg`, hsg/models/predictions/hsg.py:16-259)."""

  def __init__(self, config):
    super(Hsg, self).__init__()
    t = config.train
    self.img_sim_loss = self._construct_loss(t.img_sim_loss_types,
                                             concentration=t.img_sim_concentration)
    self.img_sim_loss_weight = t.img_sim_loss_weight
    self.fine_hrchy_loss = self._construct_loss(t.fine_hrchy_loss_types,
                                                concentration=t.fine_hrchy_concentration)
    self.fine_hrchy_loss_weight = t.fine_hrchy_loss_weight
    self.coarse_hrchy_loss = self._construct_loss(t.coarse_hrchy_loss_types,
                                                  concentration=t.coarse_hrchy_concentration)
    self.coarse_hrchy_loss_weight = t.coarse_hrchy_loss_weight
    self.dmon_loss = self._construct_loss(t.dmon_loss_types, adj_knn=t.dmon_knn)
    self.dmon_loss_weight = t.dmon_loss_weight
    self.centroid_cont_loss = self._construct_loss(t.centroid_cont_loss_types,
                                                   concentration=t.centroid_cont_concentration)
    self.centroid_cont_loss_weight = t.centroid_cont_loss_weight
    self.semantic_ignore_index = config.dataset.semantic_ignore_index
    self.num_classes = config.dataset.num_classes
    self.label_divisor = config.network.label_divisor

  _construct_loss = _construct_loss
  losses = losses

  def predictions(self, datas, targets={}):
    raise NotImplementedError()                                                        # :73-76

  def forward(self, datas, targets=None, with_loss=True, with_prediction=False):
    """:231-259."""
    targets = targets if targets is not None else {}
    outputs = {}
    if with_prediction:
      semantic_pred, semantic_score = self.predictions(datas, targets)
      outputs.update({'semantic_prediction': semantic_pred, 'semantic_score': semantic_score})
    if with_loss:
      img_sim_loss, hrchy_group_loss, clustering_loss, img_sim_acc = self.losses(datas, targets)
      accs = [a for a in [img_sim_acc] if a is not None]
      acc = sum(accs) / (len(accs) + 1e-12)
      outputs.update({'img_sim_loss': img_sim_loss, 'hrchy_group_loss': hrchy_group_loss,
                      'clustering_loss': clustering_loss, 'accuracy': acc})
    return outputs

  def get_params_lr(self):
    return []


def hsg(config):
  """Non-parametric prototype predictor (reference :270-273)."""
  return Hsg(config)


class HsgCs(Hsg):
  """`hsg/models/predictions/hsg_cs.py`: the same losses; the DMon k-NN graph is not restricted to the nodes of
  one view (`hsg_cs.py:173-175` passes no segment labels)."""
  dmon_graph_per_view = False


def hsg_cs(config):
  """Reference hsg_cs.py:269-272."""
  return HsgCs(config)
