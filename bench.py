#!/usr/bin/env python3
"""Headline benchmark: pixel-embeddings clustered per second (BASELINE.json).

One "step" = one segment_by_kmeans-equivalent pass (NCHW f32 in -> normalise ->
+location -> 10 Lloyd iterations from the grid seed -> 5 output tensors) over a
synthetic batch already resident in HBM.  N=1 workload = BASELINE.json
configs[1] (VOC12 stage-2 shape: 48x256x448x448, K=8x8).  N>1: every rank
clusters its own shard of the same per-GPU shape (weak scaling; images are
independent, k-means itself has no collective); the batch-wide prototype-table
exchange (one all_gather + one RCCL all_reduce over xGMI) is timed separately
and reported as `exchange_ms`.

Launch: `python bench.py --gpus N --steps K --warmup W`.  Under torchrun
(`WORLD_SIZE` set) the script is one rank; otherwise, for N > 1, it re-launches
itself as `python -m torch.distributed.run --nproc-per-node N ...` on
127.0.0.1, one rank per GPU.

Inputs come from the repo's portable integer-hash generator (hsg_amd/utils/synth.py,
seed 0x48534700 + cfg id, global image index = rank * B + b), generated directly in HBM
by libhsgk with the same bits numpy produces.

Prints ONE JSON line on rank 0, including
  roofline      the dominant launch group (the E-step).  `achieved` / `frac` price the HBM
                bytes the group actually moves (PMC counters of profiles/, else the bytes it
                streams by construction) over its average duration, measured with HIP events
                on the launch stream inside the timed region (libhsgk's event profiler);
                `algorithmic_bytes_over_time_GBps` is SURVEY 8(d)'s 4D+8 B per pixel over the same
                time (not a roofline fraction: the filter level reads a half-size copy of the rows);
                roofline_mstep / roofline_prep / roofline_iteration: M-step update, prep
                kernel and one whole Lloyd iteration;
  cpu_baseline  oracle/torch_ref.py (same ATen op sequence as the reference's
                CPU path) timed on the host cores, rank 0, N=1 only: warm-up + median of 3
                on 4 images, plus a 1-thread row;
  extra_runs    the same call on the 'mixture' input and with an over-segmentation label
                map + 4-row ignore band, N=1 only.
"""
import argparse
import gc
import json
import os
import socket
import statistics
import subprocess
import sys
import time

if os.environ.get('HSGK_BENCH_NO_IPC_DEFAULT') != '1':      # (set by the one-shot retry of main() after a failed RCCL start-up)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL across processes needs it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (cfg id, B per GPU, C, H, W, grid, iterations)
    'cfg2': (2, 48, 256, 448, 448, (8, 8), 10),
    'cfg3': (3, 16, 256, 224, 224, (8, 8), 10),
    'cfg5': (5, 24, 384, 224, 224, (8, 16), 10),
    'cfg1': (1, 4, 32, 64, 64, (2, 4), 10),
    'cfg4': (4, 4, 256, 768, 768, (16, 16), 10),      # finest level of the 256/64/16 hierarchy, 4 of 16 images per GPU
    # the reference's training resolution (input / 16): fixed-cost-bound small maps
    'train28': (6, 48, 256, 28, 28, (8, 8), 10),
    'train14': (7, 16, 256, 14, 14, (4, 4), 10),
    # the reference's own training hyper-parameters (bashscripts/*/train.sh: 128 channels, 4 x 4 clusters,
    # 15 iterations, 448^2 crops at output stride 8, a few images per GPU)
    'reftrain': (8, 4, 128, 56, 56, (4, 4), 15),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec


def pmc_files(workload):
  """Committed counter passes of one workload, newest round first (tools/collect_profiles_rNN.sh)."""
  names = ['r%02d_%s_pmc.txt' % (r, workload) for r in range(9, 2, -1)]
  if workload == 'cfg2':
    names += ['r02_pmc.txt', 'r01_pmc.txt']
  return names


# launch groups by what the kernels are NAMED in the library (the E-step kernels are all `assign_*`, plus the table's
# rounding-error pass; the M-step `update_sums*` / `m0_reduce*` / `xk_sums*`; the prep kernel `prep_*`), so that a
# renamed or new filter kernel cannot leave a stale list here
PMC_GROUPS = {'assign': ('assign_', 'centroid_half_err'), 'accumulate': ('update_sums', 'm0_reduce', 'xk_sums'),
              'prep': ('prep_',)}


def pmc_traffic(group, workload):
  """HBM bytes of one launch group from the committed rocprofv3 counter passes of `workload`
  (profiles/rNN_<workload>_pmc.txt: one --pmc pass per counter group, means per dispatch and dispatch counts),
  corrected as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) tallies 128-byte requests at 64 B, so
  it is doubled; WRITE_SIZE (KB) is taken as is.  The group's kernels are found by name prefix (PMC_GROUPS);
  bytes per group instance = sum over its kernels of mean x dispatches / dispatches of the group's most frequent
  kernel (one instance = one E-step of one Lloyd iteration, one M-step launch, one prep launch).
  Returns (bytes, source, kernels, commit) or (None, None, None, None)."""
  import re
  for fname in pmc_files(workload):
    path = os.path.join(ROOT, 'profiles', fname)
    try:
      lines = open(path).read().splitlines()
    except OSError:
      continue
    vals, kern, commit = {}, None, None
    for ln in lines:
      if ln.startswith('#'):
        m = re.search(r'commit ([0-9a-f]{7,40})', ln)
        if m:
          commit = m.group(1)
        continue
      if ln and not ln.startswith(' '):
        kern = ln.strip()
        continue
      parts = ln.split()
      if len(parts) >= 2 and parts[0] in ('FETCH_SIZE', 'WRITE_SIZE') and kern:
        m = re.search(r'n=(\d+)', ln)
        vals.setdefault(kern, {})[parts[0]] = (float(parts[1]) * 1024.0, int(m.group(1)) if m else 1)   # KB -> B
    hits = {k: v for k, v in vals.items() if any(pre in k for pre in PMC_GROUPS[group])}
    hits = {k: v for k, v in hits.items() if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v}
    if not hits:
      continue
    # (an E-step instance launches every kernel of its group; an M-step instance ONE of its group's kernels)
    counts = [v['FETCH_SIZE'][1] for v in hits.values()]
    n_ref = sum(counts) if group == 'accumulate' else max(counts)
    total = 0.0
    for v in hits.values():
      total += (2.0 * v['FETCH_SIZE'][0] * v['FETCH_SIZE'][1] + v['WRITE_SIZE'][0] * v['WRITE_SIZE'][1]) / n_ref
    return int(total), 'profiles/' + fname, sorted(hits), commit
  return None, None, None, None


RESULT_FD = 1          # where the JSON line goes (main() sets the real stdout aside for it)


def write_result(line):
  os.write(RESULT_FD, (json.dumps(line) + '\n').encode())


def free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def relaunch_as_ranks(n, env=None):
  """`python bench.py --gpus N` outside torchrun: start N ranks of this script on this node."""
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
         '--master-addr', '127.0.0.1', '--master-port', str(free_port()),
         os.path.abspath(__file__)] + sys.argv[1:]
  return subprocess.call(cmd, env=env)


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=3)
  ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
  ap.add_argument('--flavour', default='iid', choices=['iid', 'mixture'],
                  help='input distribution of the headline run (BASELINE.md: i.i.d. N(0,1))')
  ap.add_argument('--labels', action='store_true',
                  help='headline run with the over-segmentation label map + ignore band')
  ap.add_argument('--no-exchange', action='store_true',
                  help='skip the prototype-table exchange measurement')
  ap.add_argument('--no-extra', action='store_true', help='skip the mixture / labelled side runs')
  ap.add_argument('--dry-ranks', type=int, default=0,
                  help='N ranks on ONE device (gloo transport on device tensors; RCCL refuses several ranks per '
                       'device): exercises the rank bookkeeping, not the interconnect -- the value is not a benchmark')
  ap.add_argument('--cpu-images', type=int, default=4,
                  help='images of the workload shape timed on the host CPU (0 = skip)')
  return ap.parse_args()


def cpu_baseline(torch, x_cpu, grid, iters, shape_note):
  """oracle/torch_ref.py on the host cores: thread counts 1 and
  min(32, cores) (and min(64, cores) when the host has more: 256 oversubscribed MKL threads
  ran 45x slower than 32 in round 1, so all-cores is not tried beyond 64), one warm-up,
  median of 3 runs each; the 1-thread row uses one image."""
  from oracle import torch_ref
  nb, C, H, W = x_cpu.shape
  ncpu = os.cpu_count() or 1
  torch_ref.segment_by_kmeans(x_cpu[:1, :, :min(H, 64), :min(W, 64)].contiguous(), None, grid, None, None, 2)
  rows = []
  for threads in sorted({1, min(32, ncpu), min(64, ncpu)}):
    torch.set_num_threads(threads)
    xs = x_cpu[:1] if threads == 1 else x_cpu
    runs = []
    budget_t0 = time.perf_counter()
    for i in range(4):                       # run 0 = warm-up
      t0 = time.perf_counter()
      torch_ref.segment_by_kmeans(xs, None, grid, None, None, iters)
      dt = time.perf_counter() - t0
      if i:
        runs.append(dt)
      if time.perf_counter() - budget_t0 > 25.0 and runs:
        break
    med = statistics.median(runs)
    rows.append({'threads': threads, 'images': int(xs.shape[0]), 'median_s': round(med, 3),
                 'runs_s': [round(r, 3) for r in runs],
                 'pixels_per_s': round(xs.shape[0] * H * W / med, 1)})
  best = max(rows, key=lambda r: r['pixels_per_s'])
  one = [r for r in rows if r['threads'] == 1][0]
  return {'value': best['pixels_per_s'], 'unit': 'pixels/s', 'cores': best['threads'], 'kind': 'port',
          'single_thread_value': one['pixels_per_s'],
          'sample': '%d of the workload\'s images (%s), %d iterations, oracle/torch_ref.py (ATen op sequence '
                    'of the reference CPU path), warm-up + median of up to 3 runs per thread count; the '
                    'reference loops over images serially, so the rate does not depend on the batch size; '
                    'host has %d logical CPUs' % (nb, shape_note, iters, ncpu),
          'rows': rows}


def hierarchy_run(torch, dev, out, B, C, K):
  """cfg4's 256 -> 64 -> 16 hierarchy on the k-means output of the per-GPU images (BASELINE configs[3];
  resnet_fcn_hsg.py:228-267 of the reference without its transformer stacks, whose logits are random here):
  padded per-image prototypes, fine / coarse assignment, position-prototype group means, pixel-wise fine and
  coarse ids.  ms per call, best of 5 after a warm-up."""
  from hsg_amd.models.embeddings import hierarchy as hz
  emb, _, lab, cidx, bidx = out
  KF, KC = 64, 16
  gen = torch.Generator(device=dev)
  gen.manual_seed(11)
  fl = torch.randn((B, KF, K), device=dev, generator=gen)
  cl = torch.randn((B, KC, KF), device=dev, generator=gen)
  pos = torch.randn((emb.shape[0], C), device=dev, generator=gen)

  def best(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
      t0 = time.perf_counter()
      r = fn()
      torch.cuda.synchronize()
      ts.append(time.perf_counter() - t0)
    return round(min(ts) * 1e3, 3), r
  res = {'shape': 'B=%d images, %d pixels, nodes M=%d, fine %d, coarse %d, C=%d' % (B, emb.shape[0], K, KF, KC, C)}
  res['kmeans_prototypes_ms'], protos = best(lambda: hz.calculate_kmeans_prototypes(
      emb, cidx, bidx, pos, lab, None, label_divisor=2048, max_num_clusters=K))
  prototypes, pos_prototypes, masks, _, _, by_image = protos
  res['hier_assign_ms'], (flab, _, clab, _) = best(lambda: hz.hierarchical_grouping_from_logits(fl, cl))
  res['group_mean_fine_ms'], fine_pos = best(lambda: hz.collect_nd_coarser_prototype(
      pos_prototypes, flab, masks, num_groups=KF, normalized=False))
  hz.vouch_pixel_images(by_image, bidx)       # (as generate_clusters does: `bidx` is the id vector the tables were built from)
  res['gather_labels_fine_ms'], _ = best(lambda: hz.collect_pixel_hierarchical_clustering_indices(by_image, bidx, flab))
  res['gather_labels_coarse_ms'], _ = best(lambda: hz.collect_pixel_hierarchical_clustering_indices(by_image, bidx, clab))
  res['hierarchy_ms'] = round(sum(v for k, v in res.items() if k.endswith('_ms')), 3)
  return res


def main():
  args = parse_args()
  if args.dry_ranks > 1 and 'WORLD_SIZE' not in os.environ:
    env = dict(os.environ)
    env['HSGK_BENCH_DEVICE'] = env.get('HSGK_BENCH_DEVICE', '0')
    env['HSGK_BENCH_BACKEND'] = 'gloo'
    sys.exit(relaunch_as_ranks(args.dry_ranks, env))
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(relaunch_as_ranks(args.gpus))

  # The JSON line must be the ONLY thing on stdout: RCCL prints its version banner to the C-level stdout, buffered
  # until the process exits -- i.e. AFTER the line.  The real stdout is set aside for the line; fd 1 becomes stderr
  # for everything else (libraries, Python prints).  The descriptor survives the start-up retries (execve).
  global RESULT_FD
  if os.environ.get('HSGK_BENCH_RESULT_FD'):
    RESULT_FD = int(os.environ['HSGK_BENCH_RESULT_FD'])
  else:
    RESULT_FD = os.dup(1)
    os.set_inheritable(RESULT_FD, True)
    os.environ['HSGK_BENCH_RESULT_FD'] = str(RESULT_FD)
    sys.stdout.flush()
    os.dup2(2, 1)
  import torch
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  if 'HSGK_BENCH_DEVICE' in os.environ:            # testing aid: every rank on one device (a 1-GPU box)
    local = int(os.environ['HSGK_BENCH_DEVICE'])
  dist = None
  if world > 1 or os.environ.get('HSGK_BENCH_FORCE_DIST') == '1':   # (the switch exercises the RCCL path on one GPU)
    import datetime
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(local)
    backend = os.environ.get('HSGK_BENCH_BACKEND', 'nccl')

    def startup_failed(e, from_thread=False):
      # One retry in a fresh process image with the other IPC mode (the variable is read when the HSA runtime
      # starts, so it cannot be changed in place).  Every rank that fails does the same; a rank that did not fail
      # sits in the collective until its watchdog below sends it after the others.  The line reports which setting
      # worked.
      what = ('%s: %s' % (type(e).__name__, str(e)))[:300]
      if os.environ.get('HSGK_BENCH_RETRIED') is None and backend == 'nccl' and world > 1:
        sys.stderr.write('[bench rank %d] RCCL start-up failed (%s); retrying without HSA_ENABLE_IPC_MODE_LEGACY\n'
                         % (rank, what[:200]))
        sys.stderr.flush()
        env = dict(os.environ)
        env['HSGK_BENCH_RETRIED'] = '1'
        env['HSGK_BENCH_FIRST_ERROR'] = what
        env.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)
        env['HSGK_BENCH_NO_IPC_DEFAULT'] = '1'
        os.execve(sys.executable, [sys.executable] + sys.argv, env)
      # Both RCCL start-ups failed: the k-means path itself has no collective (images shard across the ranks), so the
      # metric is still measurable -- barrier and max over ranks through gloo; the line then says `dist_backend: gloo`,
      # `rccl_ranks: 0` and carries both RCCL errors, and the prototype exchange is timed over gloo.
      if os.environ.get('HSGK_BENCH_RETRIED') == '1' and backend == 'nccl' and world > 1:
        sys.stderr.write('[bench rank %d] RCCL start-up failed again (%s); barrier / max over gloo\n'
                         % (rank, what[:200]))
        sys.stderr.flush()
        env = dict(os.environ)
        env['HSGK_BENCH_RETRIED'] = '2'
        env['HSGK_BENCH_SECOND_ERROR'] = what
        env['HSGK_BENCH_BACKEND'] = 'gloo'
        os.execve(sys.executable, [sys.executable] + sys.argv, env)
      if rank == 0:
        write_result({'metric': 'pixel-embeddings clustered/sec', 'value': None, 'unit': 'pixels/s',
                      'n_gpus': world, 'error': 'process-group start-up failed: %s' % what,
                      'first_error': os.environ.get('HSGK_BENCH_FIRST_ERROR'),
                      'second_error': os.environ.get('HSGK_BENCH_SECOND_ERROR')})
      if from_thread:
        os._exit(1)
      sys.exit(1)

    # A rank whose peers have failed (and left for the retry) hangs in its first collective until RCCL's own
    # time-out, which ends the process instead of raising: this watchdog sends it after the others well before the
    # rendezvous of the retry (300 s) gives up on it.
    import threading
    started = threading.Event()

    def startup_watchdog():
      if not started.wait(float(os.environ.get('HSGK_BENCH_STARTUP_S', '200'))):
        startup_failed(TimeoutError('the first collective did not finish in time'), from_thread=True)

    if world > 1:
      threading.Thread(target=startup_watchdog, daemon=True).start()
    try:
      if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', local),
                                timeout=datetime.timedelta(seconds=300))
      else:
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
      # the first collective sets up the RCCL channels (IPC handles between the ranks' buffers): it is where a
      # wrong HSA IPC mode shows (`hipIpcGetMemHandle: invalid argument`)
      probe = torch.ones((1,), device=torch.device('cuda', local))
      dist.all_reduce(probe)
      torch.cuda.synchronize(torch.device('cuda', local))
      if int(probe.item()) != world:
        raise RuntimeError('first all_reduce returned %r for %d ranks' % (probe.item(), world))
      if backend == 'nccl' and os.environ.get('HSGK_BENCH_FAKE_RCCL_FAILURE') == '1':     # (fault injection, tests)
        raise RuntimeError('injected RCCL start-up failure')
      started.set()
    except Exception as e:                      # noqa: BLE001
      started.set()
      startup_failed(e)
  dev = torch.device('cuda', local)
  torch.cuda.set_device(dev)

  from hsg_amd import _lib
  from hsg_amd.utils import synth
  from hsg_amd.utils.segsort import common as segsort_common

  cfg_id, B, C, H, W, grid, iters = WORKLOADS[args.workload]
  D = C + 2
  K = grid[0] * grid[1]
  seed = synth.SEED_BASE + cfg_id
  x = synth.device_embeddings_nchw(seed, (B, C, H, W), args.flavour, dev, first_image=rank * B)

  def make_labels():
    lab = synth.overseg_labels(seed + 0x100 + rank, B, H, W, regions=48, ignore_rows=4, ignore_index=255)
    return torch.from_numpy(lab).to(dev)

  labels = make_labels() if args.labels else None

  def run(xin, lab):
    if lab is None:
      return segsort_common.segment_by_kmeans(xin, None, list(grid), iterations=iters)
    return segsort_common.segment_by_kmeans(xin, lab, list(grid), ignore_index=255, iterations=iters)

  def fence():
    torch.cuda.synchronize(dev)
    if dist is not None:
      dist.barrier()
      torch.cuda.synchronize(dev)

  def timed(xin, lab, warmup, steps):
    out = None
    gc.collect()              # (before the warm-up, which also re-warms the caches the collector's heap walk evicted)
    gc.disable()
    for _ in range(warmup):
      out = run(xin, lab)
      del out
    fence()
    t0 = time.perf_counter()
    out = None
    for _ in range(steps):
      del out
      out = run(xin, lab)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    return dt, out

  profiled = args.workload.startswith('cfg')
  # No cyclic-GC pass inside a timed region (as timeit does): a generation-2 collection of a torch process is a
  # 34-74 ms host stall that landed in about one run of eight (tools/probes/bench_stall.sh: the stalled step is a
  # CPU-side one, at a random step, gone with the collector off) -- a quarter of cfg3's ten steps.  Collected BEFORE
  # the warm-up: the collector's walk over the heap leaves the caches cold, and the first step after it paid 0.3-0.4
  # ms of host time for that (a tenth of ten 0.3 ms training-resolution steps).
  gc.collect()
  gc.disable()
  for i in range(args.warmup):
    # the LAST warm-up step already runs with the in-library events on: the runtime's first timed event record on a
    # stream is a one-time set-up (measured 9 ... 60 ms, in about one process of four) that belongs to the warm-up,
    # not to step 1 of the timed region; its records are discarded by the profile_collect() below
    if profiled and i == args.warmup - 1:
      _lib.profile_enable(True)
    out = run(x, labels)
    del out
  fence()
  # in-library HIP events around the kernel groups (the rooflines below); not for the training-resolution
  # workloads (train28 / train14 / reftrain), where ten event pairs per call would be a tenth of the 0.3 ms
  # step and no roofline is reported
  _lib.profile_enable(profiled)
  _lib.profile_collect()
  step_times = [] if os.environ.get('HSGK_BENCH_STEP_TIMES') else None    # (diagnostic: a fence per step, stderr)
  if os.environ.get('HSGK_BENCH_NOPROF'):
    _lib.profile_enable(False)
  t0 = time.perf_counter()
  out = None
  for _ in range(args.steps):
    del out
    out = run(x, labels)
    if step_times is not None:
      if os.environ['HSGK_BENCH_STEP_TIMES'] != 'nosync':
        torch.cuda.synchronize(dev)
      step_times.append(time.perf_counter())
  torch.cuda.synchronize(dev)
  if step_times:
    print('step ms:', ' '.join('%.2f' % (1e3 * (b - a)) for a, b in zip([t0] + step_times, step_times)),
          file=sys.stderr)
  own_elapsed = time.perf_counter() - t0       # this rank's own K steps (before it waits for the others)
  fence()
  elapsed = time.perf_counter() - t0
  gc.enable()
  prof = _lib.profile_collect()
  _lib.profile_enable(False)

  # The batch-wide prototype table of the step's output (local segment sums, one all_gather of
  # the segment keys and ONE RCCL all_reduce over xGMI when N > 1; hsg_amd/models/utils.py).
  # Not part of `value` (BASELINE.md metric = the segment_by_kmeans call); timed on its own
  # with the same fences, max over ranks.
  exch, exch_lib = None, None
  backend_name = (dist.get_backend() if dist is not None else None)

  def measure_exchange(library_comm):
    from hsg_amd.models import utils as model_utils
    model_utils.use_library_comm = library_comm
    try:
      emb, emb_loc, lab, cidx, bidx = out
      zeros = torch.zeros_like(lab)
      times, ncoll, res = [], 0, None
      for _ in range(4):                    # run 0 = warm-up (RCCL channel setup)
        del res
        fence()
        c0 = model_utils.collective_calls
        t1 = time.perf_counter()
        res = model_utils.gather_clustering_and_update_prototypes(emb, emb_loc, cidx, bidx, lab, zeros)
        fence()
        times.append(time.perf_counter() - t1)
        ncoll = model_utils.collective_calls - c0
      tt = torch.tensor(times[1:], device=dev, dtype=torch.float64)
      if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
      tl = tt.tolist()
      # every rank must hold the SAME batch-wide tables: a bit-level checksum of both float tables and the three
      # label vectors, gathered from all ranks
      chk = torch.stack([res[0].contiguous().view(torch.int32).long().sum(), res[1].contiguous().view(torch.int32).long().sum(),
                         res[2].sum(), res[3].sum(), res[4].sum(),
                         torch.tensor(res[0].shape[0], device=dev)]).to(torch.int64)
      allchk = [chk]
      if dist is not None:
        allchk = [torch.empty_like(chk) for _ in range(world)]
        dist.all_gather(allchk, chk)
      same = all(bool(torch.equal(c, allchk[0])) for c in allchk)
      return {'ms': round(statistics.median(tl) * 1e3, 3), 'runs_ms': [round(t * 1e3, 3) for t in tl],
              'table_checksum': [int(v) for v in allchk[0].tolist()], 'table_checksum_equal_on_all_ranks': same,
              'ranks_compared': len(allchk),
              'segments_total': int(res[0].shape[0]), 'collectives_per_call': int(ncoll),
              'payload_MB': round(res[0].shape[0] * (2 * C + 2) * 4 / 1e6, 2),
              'transport': ('libhsgk RCCL communicator, in-stream' if library_comm else
                            'torch.distributed (%s)' % backend_name) if world > 1 else 'none (one rank)'}
    finally:
      model_utils.use_library_comm = False

  if not args.no_exchange:
    try:        # never let the side measurement take the headline line down with it
      exch = measure_exchange(False)
    except Exception as e:                      # noqa: BLE001
      exch = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
  if not (world > 1 and backend_name == 'nccl'):
    out = None                                  # (kept for the in-stream variant at the end otherwise)
  own_rates = [B * H * W * args.steps / own_elapsed]
  if dist is not None:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    mine = torch.tensor(own_rates, device=dev, dtype=torch.float64)
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    own_rates = [float(v.item()) for v in every]

  px_per_step = B * H * W * world
  value = px_per_step * args.steps / elapsed

  # Rooflines (HBM-bound kernels; durations from libhsgk's HIP-event profiler, recorded on
  # the launch stream inside the timed region).  `roofline` is the DOMINANT launch group by
  # time, the E-step (also the one north_star's 50 % target names).
  npx = B * H * W
  p_ms, p_n = prof['prep']
  m_ms, m_n = prof['accumulate']
  f_ms, f_n = prof['finalize']
  a_ms, a_n = prof['assign']
  headline = args.workload == 'cfg2' and B == 48 and args.flavour == 'iid' and not args.labels

  def rl(kernel, algorithmic, moved, moved_source, ms, n, commit=None, **extra):
    """achieved / frac: the bytes the launch group moves through HBM (committed PMC counters of this workload, or a
    by-construction count) over its duration measured in THIS run; both are null when neither source exists --
    SURVEY 8(d)'s algorithmic bytes over the same time are reported next to them but are not a bandwidth (the
    filter level reads a half-size copy, the update kernel only the changed rows)."""
    if not n:
      return None
    avg_s = ms / n * 1e-3
    alg = algorithmic / avg_s / 1e9
    ach = moved / avg_s / 1e9 if moved else None
    res = {'bound': 'hbm', 'kernel': kernel, 'achieved': round(ach, 1) if ach else None, 'peak': HBM_PEAK_GBS,
           'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4) if ach else None,
           'traffic': int(moved) if moved and moved_source.startswith('profiles/') else None,
           'bytes_source': moved_source, 'traffic_measured_in_this_run': False, 'traffic_counters_commit': commit,
           'avg_launch_ms': round(ms / n, 4), 'launches': int(n),
           'algorithmic_bytes_per_launch': int(algorithmic),
           'algorithmic_bytes_over_time_GBps': round(alg, 1)}
    res.update(extra)
    return res

  # counters apply to the workload's own per-GPU batch on the i.i.d. input without a label map (what was profiled)
  profiled = args.flavour == 'iid' and not args.labels and B == WORKLOADS[args.workload][1]
  half_ok = (D % 64 == 2 and 128 <= D <= 322 and (D // 64) % 2 == 0 and K <= 64)
  e_alg = (4 * D + 8) * npx
  e_moved, e_src, e_names, e_commit = None, 'no counter pass committed for this workload / input', None, None
  if profiled:
    tr, src, names, commit = pmc_traffic('assign', args.workload)
    if tr:
      e_moved, e_names, e_commit = tr, names, commit
      e_src = src + ': 2 x FETCH_SIZE + WRITE_SIZE summed over the group (gfx950 correction of MI355X_MICROARCH.md)'
  if e_moved is None and half_ok:
    e_moved = (2 * (D - 2) + 8 + 4 + 0.015 * 4 * D) * npx
    e_src = 'by construction: fp16 row copy 2(D-2)+8 B + 4 B label per pixel + the fp32 rows of ~1.5 % undecided pixels'
  roofline = rl(
      'E-step launch group (dominant by time)' + (': ' + ' + '.join(e_names) if e_names else ''),
      e_alg, e_moved, e_src, a_ms, a_n, e_commit,
      mfma_tflops=round(2.0 * D * K * npx / (a_ms / max(a_n, 1) * 1e-3) / 1e12, 2) if a_n else None)

  m_alg = 4 * D * npx
  m_moved, m_src, m_commit = None, 'no counter pass committed for this workload / input', None
  pr_alg = (8 * C + 4 * D + 24) * npx
  pr_moved, pr_src, pr_commit = pr_alg + (2 * C + 12) * npx, 'by construction: input + both float outputs + labels + the fp16 row copy', None
  if profiled:
    t_m, s_m, _, c_m = pmc_traffic('accumulate', args.workload)
    if t_m:        # (mean over the call's M-step launches: one m0_reduce / first update + (iterations - 1) updates)
      m_moved, m_src, m_commit = t_m, s_m + ': 2 x FETCH_SIZE + WRITE_SIZE, mean over the call\'s M-step launches', c_m
    t_p, s_p, _, c_p = pmc_traffic('prep', args.workload)
    if t_p:
      pr_moved, pr_src, pr_commit = t_p, s_p + ': 2 x FETCH_SIZE + WRITE_SIZE', c_p
  roofline_mstep = rl(
      'M-step: exact fixed-point segment sums.  The first M-step of a call is summed inside the prep '
      'kernel (seed-grid labels) and folded by m0_reduce_kernel where the prep kernel can; the other launches are the '
      'update_sums kernel, which reads only the rows whose label changed', m_alg, m_moved, m_src, m_ms, m_n, m_commit)
  roofline_prep = rl('prep kernel (NCHW -> normalised rows, both float outputs, labels, fp16 copy, first M-step)',
                     pr_alg, pr_moved, pr_src, p_ms, p_n, pr_commit)
  roofline_iteration = None
  if m_n and a_n and f_n:
    it_ms = m_ms / m_n + f_ms / f_n + a_ms / a_n
    roofline_iteration = rl('one Lloyd iteration = sums update + finalize + E-step group', e_alg,
                            (e_moved + m_moved) if (e_moved and m_moved) else None,
                            'sum of the E-step and M-step figures above', it_ms, 1, e_commit)
    roofline_iteration['launches'] = int(a_n)
  phases = {k: round(v[0] / max(1, args.steps), 3) for k, v in prof.items()}

  # Side runs on the other BASELINE.md inputs (N=1 only; 1 warm-up + 2 timed calls each).
  extra = None
  if rank == 0 and world == 1 and not args.no_extra:
    extra = {}
    try:
      xm = synth.device_embeddings_nchw(seed, (B, C, H, W), 'mixture' if args.flavour == 'iid' else 'iid', dev)
      dt, o = timed(xm, None, 1, 2)
      extra['mixture' if args.flavour == 'iid' else 'iid'] = {
          'ms_per_step': round(dt / 2 * 1e3, 3), 'pixels_per_s': round(npx * 2 / dt, 1)}
      del o, xm
      lab2 = make_labels()
      dt, o = timed(x, lab2, 1, 2)
      extra['iid_overseg_labels_ignore_band'] = {
          'ms_per_step': round(dt / 2 * 1e3, 3), 'pixels_per_s': round(npx * 2 / dt, 1),
          'kept_pixels': int(o[0].shape[0]), 'segments': int(o[3].max()) + 1}
      del o, lab2
    except Exception as e:                      # noqa: BLE001
      extra['error'] = '%s: %s' % (type(e).__name__, str(e)[:200])
    if args.workload == 'cfg4':
      try:
        extra['hierarchy'] = hierarchy_run(torch, dev, timed(x, None, 0, 1)[1], B, C, K)
      except Exception as e:                    # noqa: BLE001
        extra['hierarchy_error'] = '%s: %s' % (type(e).__name__, str(e)[:200])

  cpu = None
  if rank == 0 and world == 1 and args.cpu_images > 0:
    nb = min(args.cpu_images, B)
    cpu = cpu_baseline(torch, x[:nb].cpu(), grid, iters, '%dx%dx%d each' % (C, H, W))

  def emit(instream):
    if rank != 0:
      return
    write_result({
        'metric': 'pixel-embeddings clustered/sec', 'value': round(value, 1),
        'unit': 'pixels/s', 'n_gpus': 1 if args.dry_ranks > 1 else world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: segment_by_kmeans %dx%dx%dx%d per GPU, K=%dx%d, %d Lloyd '
                               'iterations, %s input (portable generator, seed 0x%X), %s'
                               % (args.workload, B, C, H, W, grid[0], grid[1], iters, args.flavour, seed,
                                  'over-segmentation labels + ignore band' if args.labels else 'no labels'),
                   'per_gpu_batch': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world,
                   'phase_ms_per_step': phases},
        'exchange_ms': exch.get('ms') if exch else None, 'prototype_exchange': exch,
        'prototype_exchange_instream': instream, 'dist_backend': backend_name,
        'rccl_ranks': world if backend_name == 'nccl' else 0, 'dry_ranks': args.dry_ranks if args.dry_ranks > 1 else 0,
        # rank 0's own rate over its K steps before the closing barrier (at N = 1 it equals `value`), and every rank's
        'n1_value': round(own_rates[0], 1), 'per_rank_pixels_per_s': [round(v, 1) for v in own_rates],
        'hsa_ipc_mode_legacy_env': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'),
        'rccl_startup_retried': os.environ.get('HSGK_BENCH_RETRIED') in ('1', '2'),
        'rccl_startup_first_error': os.environ.get('HSGK_BENCH_FIRST_ERROR'),
        'rccl_startup_second_error': os.environ.get('HSGK_BENCH_SECOND_ERROR'),
        'rccl_fell_back_to_gloo': os.environ.get('HSGK_BENCH_RETRIED') == '2',
        'roofline': roofline, 'roofline_mstep': roofline_mstep, 'roofline_prep': roofline_prep,
        'roofline_iteration': roofline_iteration, 'cpu_baseline': cpu, 'extra_runs': extra})

  # The same exchange with both collectives IN-STREAM on libhsgk's own RCCL communicator (the C entry points
  # hsgk_exchange_* / hsgk_comm_*): only with real RCCL ranks.  It runs last, under a watchdog that prints the
  # line without it and ends the process if a collective should hang, so the headline line never depends on it.
  instream = None
  if world > 1 and backend_name == 'nccl' and not args.no_exchange and exch and 'error' not in exch:
    import threading
    done = threading.Event()

    def watchdog():
      if not done.wait(120.0):
        emit({'error': 'timeout: no result within 120 s'})
        os._exit(0)

    threading.Thread(target=watchdog, daemon=True).start()
    try:
      instream = measure_exchange(True)
    except Exception as e:                      # noqa: BLE001
      instream = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    done.set()
  emit(instream)
  if dist is not None:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
