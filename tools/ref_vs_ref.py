#!/usr/bin/env python3
"""Reference against ITSELF (round 6, review item 2; build container only -- imports /root/reference).

`north_star` asks for cluster assignments bit-identical to the reference's CPU path.  The reference leaves the float
summation order of its E-step to the BLAS behind `torch.mm` and of its normalisation to ATen's vectorised reduction
(hsg/utils/segsort/common.py:62-64, hsg/utils/general/common.py:116-120), so "the reference's labels" are a function
of the host the reference runs on.  This script measures by how much: the f19 images (image 0 of every BASELINE
batch, i.i.d. and mixture) through the reference's own functions under several execution settings of the SAME
reference code on the SAME machine --

    base     torch.set_num_threads(8)                      (what tools/gen_golden.py f19 ran)
    t1 / t3  1 and 3 threads
    avx2     MKL_CBWR=AVX2              (the sgemm kernels MKL picks on an AVX2-only host)
    compat   MKL_CBWR=COMPATIBLE        (MKL's ISA-independent kernels)
    aten2    ATEN_CPU_CAPABILITY=avx2   (ATen's reduction kernels of an AVX2-only host: the normalisation)

and records, per image and setting, the pixels whose label differs from `base`
  * teacher-forced: ONE reference iteration started from base's labels after iteration t - 1, against base's labels
    after iteration t (t = 1 .. 10) -- what a different summation order alone does in one step, and
  * free-running: the setting's own ten iterations against base's, per iteration,
next to the same two numbers for our canonical arithmetic (from the f19 fixtures).

    python tools/ref_vs_ref.py            -> tests/golden/f20_ref_vs_ref.npz, profiles/r06_ref_vs_ref.txt
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [            # (name, threads, extra environment)
    ('base', 8, {}),
    ('t1', 1, {}),
    ('t3', 3, {}),
    ('avx2', 8, {'MKL_CBWR': 'AVX2'}),
    ('compat', 8, {'MKL_CBWR': 'COMPATIBLE'}),
    ('aten2', 8, {'ATEN_CPU_CAPABILITY': 'avx2'}),
]
CASES = [('cfg2', 2, (256, 448, 448), (8, 8)), ('cfg3', 3, (256, 224, 224), (8, 8)),
         ('cfg4', 4, (256, 768, 768), (16, 16)), ('cfg5', 5, (384, 224, 224), (8, 16))]
TMP = '/tmp/ref_vs_ref'


def worker(threads, tag, flav, out, base):
  import torch
  torch.set_num_threads(threads)
  sys.path.insert(0, ROOT)
  sys.path.insert(0, '/root/reference')
  import hsg.utils.general.common as ref_general
  import hsg.utils.segsort.common as ref_common
  from hsg_amd.utils import synth
  _, cid, (C, H, W), grid = next(c for c in CASES if c[0] == tag)
  x = synth.embeddings_nchw(synth.SEED_BASE + cid, (1, C, H, W), flav)
  # the operator's prologue (common.py:306-352) for the one image
  e = ref_general.normalize_embedding(torch.from_numpy(x).permute(0, 2, 3, 1).contiguous())
  loc = ref_common.generate_location_features((H, W), 'cpu', 'float') - 0.5
  rows = ref_general.normalize_embedding(torch.cat([e[0].view(-1, C), loc.view(-1, 2)], -1))
  K = grid[0] * grid[1]
  init = ref_common.initialize_cluster_labels(list(grid), (H, W), 'cpu').view(-1)
  _, init = torch.unique(init, return_inverse=True)
  free = [init.numpy()]
  for it in range(1, 11):
    free.append(ref_common.kmeans_with_initial_labels(rows, torch.from_numpy(free[-1]), K, 1).numpy())
  rec = dict(free=np.stack(free).astype(np.uint8), rows=rows.numpy())
  if base:
    b = np.load(base)
    bl = b['free'].astype(np.int64)
    tf = []
    for t in range(1, 11):
      got = ref_common.kmeans_with_initial_labels(rows, torch.from_numpy(bl[t - 1]), K, 1).numpy()
      tf.append(int((got != bl[t]).sum()))
    rec['tf_counts'] = np.array(tf, np.int64)
    rec['free_counts'] = np.array([int((free[t] != bl[t]).sum()) for t in range(1, 11)], np.int64)
    rec['rows_differing'] = np.int64(int((rows.numpy() != b['rows']).sum()))
    rec['rows_max_abs'] = np.float64(np.abs(rows.numpy() - b['rows']).max())
    del rec['rows']
  np.savez(out, **rec)


def main():
  os.makedirs(TMP, exist_ok=True)
  fix = {}
  lines = []
  for tag, cid, (C, H, W), grid in CASES:
    for flav in ('iid', 'mixture'):
      key = '%s_%s' % (tag, flav)
      f19 = np.load(os.path.join(ROOT, 'tests', 'golden', 'f19_full_%s.npz' % key))
      ours_tf = f19['tf_counts'].astype(np.int64)
      ours_free = int(f19['free_pixels'].size)
      lines.append('%s  (%d pixels, K = %d)' % (key, H * W, grid[0] * grid[1]))
      lines.append('   %-8s teacher-forced per iteration %-44s free-running after 10: %d' %
                   ('ours', ' '.join('%d' % v for v in ours_tf), ours_free))
      fix[key + '_ours_tf'] = ours_tf
      fix[key + '_ours_free10'] = np.int64(ours_free)
      base = os.path.join(TMP, '%s_base.npz' % key)
      for name, threads, env in SETTINGS:
        out = os.path.join(TMP, '%s_%s.npz' % (key, name))
        e = dict(os.environ)
        e.update(env)
        if not os.path.exists(out):                  # (resumable: a finished setting of an image is kept)
          subprocess.check_call([sys.executable, __file__, 'worker', str(threads), tag, flav, out + '.tmp.npz',
                                 '' if name == 'base' else base], env=e)
          os.replace(out + '.tmp.npz', out)
        r = np.load(out)
        if name == 'base':
          # the same run as the f19 fixture's (labels after iterations 1, 2, 9)
          assert np.array_equal(r['free'][1], f19['lab1']) and np.array_equal(r['free'][9], f19['lab9']), key
          # the reference's labels after iterations 3 .. 8 (f19 holds 1, 2, 9, 10), each as a delta against the
          # iteration before: with them the GPU test teacher-forces ALL ten iterations (tests/util.py: f19_case)
          for t in range(3, 9):
            idx = np.nonzero(r['free'][t] != r['free'][t - 1])[0]
            fix['%s_lab%d_idx' % (key, t)] = idx.astype(np.int32)
            fix['%s_lab%d_val' % (key, t)] = r['free'][t][idx].astype(np.uint8)
          continue
        fix['%s_%s_tf' % (key, name)] = r['tf_counts']
        fix['%s_%s_free' % (key, name)] = r['free_counts']
        fix['%s_%s_rows_differing' % (key, name)] = r['rows_differing']
        lines.append('   %-8s teacher-forced per iteration %-44s free-running per iteration: %s   (row elements '
                     'differing from base: %d, max |diff| %.2g)' %
                     (name, ' '.join('%d' % v for v in r['tf_counts']), ' '.join('%d' % v for v in r['free_counts']),
                      int(r['rows_differing']), float(r['rows_max_abs'])))
      print('\n'.join(lines[-(len(SETTINGS) + 1):]), flush=True)
  fix['settings'] = np.array([s[0] for s in SETTINGS[1:]])
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'f20_ref_vs_ref.npz'), **fix)
  with open(os.path.join(ROOT, 'profiles', 'r06_ref_vs_ref.txt'), 'w') as f:
    f.write('# tools/ref_vs_ref.py: the reference (twke18/HSG, torch 2.10 CPU, MKL 2024.2) against itself under other '
            'execution settings,\n# differing pixels against its 8-thread default run (= the f19 fixtures); "ours" = the '
            'canonical arithmetic against the same run\n')
    f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == 'worker':
    worker(int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6])
  else:
    main()
