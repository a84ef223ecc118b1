"""`hsg_amd.patch_reference()`: rebinds the hot path of an importable twke18/HSG checkout to
libhsgk so that `pyscripts/train/train.py` and the inference scripts run unchanged.

What is rebound (reference path -> hsg_amd implementation):
  hsg.utils.segsort.common      segment_by_kmeans, kmeans_with_initial_labels, kmeans,
                                find_nearest_prototypes, calculate_prototypes_from_labels,
                                prepare_prototype_labels, find_majority_label_index,
                                initialize_cluster_labels, generate_location_features
  hsg.utils.general.common      normalize_embedding, segment_mean
  hsg.utils.segsort.loss        SegSortLoss, SetSegSortLoss
  hsg.utils.segsort.eval        top_k_ranking
  hsg.utils.graph.common / loss affinity_matrix_as_attention; DMonLoss, HierarchicalDMonLoss, dmon_pool_loss,
                                NCutLoss, ncut_pool_loss
  hsg.utils.segsort.others      load_memory_banks (the .npy prototype bank)
  hsg.models.utils              gather_and_reorder_image_indices, gather_and_update_cluster_mappings,
                                gather_clustering_and_update_prototypes, gather_and_update_datas,
                                gather_multiset_labels_per_batch_by_nearest_neighbor
  hsg.models.embeddings.resnet_fcn_hsg   ResnetFcn / MultiviewResnetFcn: generate_clusters,
                                _calculate_kmeans_prototypes, _hierarchical_grouping,
                                _collect_nd_coarser_prototype,
                                _collect_pixel_hierarchical_clustering_indices
  hsg.models.embeddings.resnet_fcn_hsg_cs   the Cityscapes twin: the same methods, pad length = the call's maximum
  hsg.models.embeddings.resnet_fcn   ResnetFcn.generate_clusters (stage-1 / inference model: k-means only)
  hsg.models.embeddings.transformer_clusters   TransformerClustering.forward (tail on libhsgk)
  hsg.models.predictions.hsg    Hsg._construct_loss, Hsg.losses (three losses, one E P^T pass)
  hsg.models.predictions.hsg_cs   the same two, the DMon graph over whole image rows
  hsg.models.predictions.segsort   Segsort.predictions (ONE top-k launch instead of the 10-chunk retrieval loop,
                                pyscripts/inference/inference.py:70,92), Segsort.losses (batched), _construct_loss

Modules that import the op library under an alias (`import hsg.utils.segsort.common as
segsort_common`) hold a reference to the module object, so rebinding the module's attributes
reaches them too.  The backbone, transformer stacks, data pipeline, optimiser and SyncBN are
untouched.  Returns the list of 'module.attribute' names that were rebound."""
import importlib


def _rebind(module, names, source, done):
  for name in names:
    if hasattr(module, name) and hasattr(source, name):
      setattr(module, name, getattr(source, name))
      done.append('%s.%s' % (module.__name__, name))


def patch_reference(package='hsg'):
  from hsg_amd.models import utils as mu
  from hsg_amd.models.embeddings import resnet_fcn as stage1_mirror
  from hsg_amd.models.embeddings import resnet_fcn_hsg as emb_mirror
  from hsg_amd.models.embeddings import transformer_clusters as tc_mirror
  from hsg_amd.models.predictions import hsg as pred_mirror
  from hsg_amd.models.predictions import segsort as segsort_mirror
  from hsg_amd.utils.segsort import others as so
  from hsg_amd.utils.general import common as gc
  from hsg_amd.utils.graph import common as graph_c
  from hsg_amd.utils.graph import loss as graph_l
  from hsg_amd.utils.segsort import common as sc
  from hsg_amd.utils.segsort import eval as se
  from hsg_amd.utils.segsort import loss as sl

  def mod(name):
    try:
      return importlib.import_module(package + '.' + name)
    except ImportError:
      return None

  done = []
  plan = [
      ('utils.segsort.common', sc, ['segment_by_kmeans', 'kmeans_with_initial_labels', 'kmeans',
                                    'find_nearest_prototypes', 'calculate_prototypes_from_labels',
                                    'prepare_prototype_labels', 'find_majority_label_index',
                                    'initialize_cluster_labels', 'generate_location_features']),
      ('utils.general.common', gc, ['normalize_embedding', 'segment_mean']),
      ('utils.segsort.loss', sl, ['SegSortLoss', 'SetSegSortLoss']),
      ('utils.segsort.eval', se, ['top_k_ranking']),
      ('utils.graph.common', graph_c, ['affinity_matrix_as_attention']),
      ('utils.graph.loss', graph_l, ['DMonLoss', 'HierarchicalDMonLoss', 'dmon_pool_loss', 'NCutLoss', 'ncut_pool_loss']),
      ('utils.segsort.others', so, ['load_memory_banks']),
      ('models.utils', mu, ['gather_and_reorder_image_indices', 'gather_and_update_cluster_mappings',
                            'gather_clustering_and_update_prototypes', 'gather_and_update_datas',
                            'gather_multiset_labels_per_batch_by_nearest_neighbor']),
  ]
  for name, source, attrs in plan:
    m = mod(name)
    if m is not None:
      _rebind(m, attrs, source, done)

  methods_cs = ['_calculate_kmeans_prototypes', '_hierarchical_grouping', '_collect_nd_coarser_prototype',
                '_collect_pixel_hierarchical_clustering_indices']
  m = mod('models.embeddings.resnet_fcn_hsg')
  if m is not None:
    methods = ['_calculate_kmeans_prototypes', '_hierarchical_grouping', '_collect_nd_coarser_prototype',
               '_collect_pixel_hierarchical_clustering_indices']
    for cls_name, gen in (('ResnetFcn', emb_mirror.generate_clusters),
                          ('MultiviewResnetFcn', emb_mirror.generate_clusters_multiview)):
      cls = getattr(m, cls_name, None)
      if cls is None:
        continue
      cls.generate_clusters = gen
      done.append('%s.%s.generate_clusters' % (m.__name__, cls_name))
      for name in methods:
        setattr(cls, name, getattr(emb_mirror, name))
        done.append('%s.%s.%s' % (m.__name__, cls_name, name))
  # the Cityscapes twin of the model (resnet_fcn_hsg_cs.py): the same methods, pad length = the call's maximum
  m = mod('models.embeddings.resnet_fcn_hsg_cs')
  if m is not None:
    for cls_name, gen in (('ResnetFcn', emb_mirror.generate_clusters),
                          ('MultiviewResnetFcn', emb_mirror.generate_clusters_multiview)):
      cls = getattr(m, cls_name, None)
      if cls is None:
        continue
      cls.generate_clusters = gen
      cls.dynamic_max_num_clusters = True
      done.append('%s.%s.generate_clusters' % (m.__name__, cls_name))
      for name in methods_cs:
        setattr(cls, name, getattr(emb_mirror, name))
        done.append('%s.%s.%s' % (m.__name__, cls_name, name))
  # the stage-1 / inference model (resnet_fcn.py): k-means only
  m = mod('models.embeddings.resnet_fcn')
  if m is not None and hasattr(m, 'ResnetFcn'):
    m.ResnetFcn.generate_clusters = stage1_mirror.generate_clusters
    done.append(m.__name__ + '.ResnetFcn.generate_clusters')
  m = mod('models.embeddings.transformer_clusters')
  if m is not None and hasattr(m, 'TransformerClustering'):
    m.TransformerClustering.forward = tc_mirror.forward
    done.append(m.__name__ + '.TransformerClustering.forward')
  m = mod('models.predictions.hsg')
  if m is not None and hasattr(m, 'Hsg'):
    m.Hsg._construct_loss = pred_mirror._construct_loss
    m.Hsg.losses = pred_mirror.losses
    done += [m.__name__ + '.Hsg._construct_loss', m.__name__ + '.Hsg.losses']
  m = mod('models.predictions.hsg_cs')
  if m is not None and hasattr(m, 'Hsg'):
    m.Hsg._construct_loss = pred_mirror._construct_loss
    m.Hsg.losses = pred_mirror.losses
    m.Hsg.dmon_graph_per_view = False
    done += [m.__name__ + '.Hsg._construct_loss', m.__name__ + '.Hsg.losses']
  m = mod('models.predictions.segsort')
  if m is not None and hasattr(m, 'Segsort'):
    for name in ('predictions', 'losses', '_construct_loss'):
      setattr(m.Segsort, name, getattr(segsort_mirror.Segsort, name))
      done.append('%s.Segsort.%s' % (m.__name__, name))
    for name in ('_prototype_table', '_semantic_losses', '_image_similarity_loss'):   # helpers of the mirror's losses
      setattr(m.Segsort, name, getattr(segsort_mirror.Segsort, name))
  return done
