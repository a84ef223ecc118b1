"""Drop-in for `hsg.models.predictions.segsort.Segsort` (reference
hsg/models/predictions/segsort.py): nearest-neighbour semantic prediction against a
prototype memory bank (:66-123, the inference-time classifier) and the supervised /
weakly supervised SegSort losses (:125-252), on the hsg_amd operators.

`predictions`: the reference retrieves the 20 nearest memory prototypes of every segment
in up to ten chunks (each an [n, M] similarity matrix + argsort); here it is ONE top-k
launch over all segments (csrc/topk.hip), followed by the same majority vote.
"""
import torch
import torch.nn as nn

import hsg_amd.utils.segsort.common as segsort_common
import hsg_amd.utils.segsort.eval as segsort_eval
import hsg_amd.utils.segsort.loss as segsort_loss


class Segsort(nn.Module):

  def __init__(self, config):
    super(Segsort, self).__init__()
    t = config.train
    self.sem_ann_loss = self._construct_loss(t.sem_ann_loss_types, concentration=t.sem_ann_concentration)
    self.sem_ann_loss_weight = t.sem_ann_loss_weight
    loss_type = 'set_segsort' if t.sem_occ_loss_types == 'segsort' else 'none'            # :27-32
    self.sem_occ_loss = self._construct_loss(loss_type, concentration=t.sem_occ_concentration)
    self.sem_occ_loss_weight = t.sem_occ_loss_weight
    self.img_sim_loss = self._construct_loss(t.img_sim_loss_types, concentration=t.img_sim_concentration)
    self.img_sim_loss_weight = t.img_sim_loss_weight
    self.feat_aff_loss = self._construct_loss(t.feat_aff_loss_types, concentration=t.feat_aff_concentration)
    self.feat_aff_loss_weight = t.feat_aff_loss_weight
    self.semantic_ignore_index = config.dataset.semantic_ignore_index
    self.num_classes = config.dataset.num_classes
    self.label_divisor = config.network.label_divisor

  def _construct_loss(self, loss_types, **kwargs):
    """:52-64."""
    if loss_types == 'segsort':
      return segsort_loss.SegSortLoss(kwargs['concentration'], group_mode='segsort+', reduction='mean')
    elif loss_types == 'set_segsort':
      return segsort_loss.SetSegSortLoss(kwargs['concentration'], group_mode='segsort+', reduction='mean')
    return None

  def predictions(self, datas, targets={}):
    """:66-123: (semantic_pred [num_pixels], semantic_topk [num_pixels, 20]) or (None, None)."""
    memory_prototypes = targets.get('semantic_memory_prototype', None)
    memory_prototype_labels = targets.get('semantic_memory_prototype_label', None)
    cluster_embeddings = datas.get('cluster_embedding', None)
    cluster_indices = datas.get('cluster_index', None)
    if (memory_prototypes is None or memory_prototype_labels is None or cluster_embeddings is None
        or cluster_indices is None):
      return None, None
    _, cluster_indices = torch.unique(cluster_indices, return_inverse=True)
    num_prototypes = int(cluster_indices.max()) + 1
    prototypes = segsort_common.calculate_prototypes_from_labels(cluster_embeddings, cluster_indices,
                                                                 num_prototypes)
    idx, _ = segsort_eval.top_k_indices(prototypes, memory_prototypes, 20)
    top_k_labels = memory_prototype_labels.view(-1)[idx.view(-1)].view(-1, 20)
    pred = segsort_eval.majority_label_from_topk(top_k_labels)
    return (torch.gather(pred, 0, cluster_indices), torch.index_select(top_k_labels, 0, cluster_indices))

  def losses(self, datas, targets={}):
    """:125-222: (sem_ann_loss, sem_occ_loss, img_sim_loss, sem_ann_acc)."""
    sem_ann_loss = sem_occ_loss = img_sim_loss = sem_ann_acc = None
    if self.sem_ann_loss is not None or self.sem_occ_loss is not None:
      cluster_indices = datas['cluster_index']
      embeddings = datas['cluster_embedding']
      semantic_labels = datas['cluster_semantic_label']
      batch_indices = datas['cluster_batch_index']
      prototypes = targets['prototype']
      prototype_semantic_labels = targets['prototype_semantic_label']
      prototype_batch_indices = targets['prototype_batch_index']
      semantic_tags = torch.index_select(targets['semantic_tag'][:, 1:self.num_classes], 0, batch_indices)
      prototype_semantic_tags = targets['prototype_semantic_tag'][:, 1:self.num_classes]
      mem_p = targets.get('memory_prototype', [])
      mem_l = targets.get('memory_prototype_semantic_label', [])
      mem_b = targets.get('memory_prototype_batch_index', [])
      mem_t = targets.get('memory_prototype_semantic_tag', [])
      if mem_p and mem_l and mem_t and mem_b:                                             # :156-179
        prototypes = torch.cat([prototypes] + list(mem_p), dim=0)
        prototype_semantic_labels = torch.cat([prototype_semantic_labels] + list(mem_l), dim=0)
        prototype_semantic_tags = torch.cat(
            [prototype_semantic_tags] + [lab[:, 1:self.num_classes] for lab in mem_t], dim=0)
        prototype_batch_indices = torch.cat([prototype_batch_indices] + list(mem_b), dim=0)
      pixel_inds = (semantic_labels < self.num_classes).nonzero().view(-1)
      proto_inds = (prototype_semantic_labels < self.num_classes).nonzero().view(-1)
      c_inds = torch.arange(prototypes.shape[0], dtype=torch.long, device=prototypes.device)
      c_inds = c_inds.masked_fill(prototype_semantic_labels >= self.num_classes, c_inds.max() + 1)
      _, c_inds = torch.unique(c_inds, return_inverse=True)
      new_cluster_indices = torch.gather(c_inds, 0, cluster_indices)
      sem_ann_loss = self.sem_ann_loss(
          torch.index_select(embeddings, 0, pixel_inds), torch.index_select(semantic_labels, 0, pixel_inds),
          torch.index_select(new_cluster_indices, 0, pixel_inds), torch.index_select(prototypes, 0, proto_inds),
          torch.index_select(prototype_semantic_labels, 0, proto_inds)) * self.sem_ann_loss_weight
      sem_occ_loss = self.sem_occ_loss(embeddings, semantic_tags, cluster_indices, prototypes,
                                       prototype_semantic_tags) * self.sem_occ_loss_weight
      sem_ann_acc, _ = segsort_eval.top_k_ranking(prototypes, prototype_semantic_labels, prototypes,
                                                  prototype_semantic_labels, 5)
    if self.img_sim_loss is not None:                                                     # :224-244
      cluster_indices = datas['cluster_index']
      embeddings = datas['cluster_embedding_with_loc']
      instance_labels = datas['cluster_instance_label']
      batch_indices = datas['cluster_batch_index']
      parts = []
      for batch_ind in torch.unique(batch_indices):
        inds = (batch_indices == batch_ind).nonzero().view(-1)
        embs = torch.index_select(embeddings, 0, inds)
        labs = torch.index_select(instance_labels, 0, inds)
        c_inds = torch.index_select(cluster_indices, 0, inds)
        p_labs, c_inds = segsort_common.prepare_prototype_labels(labs, c_inds, labs.max() + 1)
        protos = segsort_common.calculate_prototypes_from_labels(embs, c_inds)
        parts.append(self.img_sim_loss(embs, labs, c_inds, protos, p_labs))
      img_sim_loss = sum(parts) / len(parts) * self.img_sim_loss_weight
    return sem_ann_loss, sem_occ_loss, img_sim_loss, sem_ann_acc

  def forward(self, datas, targets=None, with_loss=True, with_prediction=False):
    """:254-279."""
    targets = targets if targets is not None else {}
    outputs = {}
    if with_prediction:
      semantic_pred, semantic_score = self.predictions(datas, targets)
      outputs.update({'semantic_prediction': semantic_pred, 'semantic_score': semantic_score})
    if with_loss:
      sem_ann_loss, sem_occ_loss, img_sim_loss, sem_ann_acc = self.losses(datas, targets)
      outputs.update({'sem_ann_loss': sem_ann_loss, 'sem_occ_loss': sem_occ_loss,
                      'img_sim_loss': img_sim_loss, 'accuracy': sem_ann_acc})
    return outputs

  def get_params_lr(self):
    return []


def segsort(config):
  """Non-parametric prototype predictor (reference :290-293)."""
  return Segsort(config)
