// torch_ops.cpp -- the torch-extension binding of libhsgk.so ("TORCH_LIBRARY(hsgk, ...) +
// autograd.Function"): host C++ only, no device code.  Every op allocates its outputs and workspace from torch's
// caching allocator, launches on the current HIP stream of the tensors' device through the C ABI of include/hsgk.h,
// (PyTorch-ROCm presents the device as type `cuda`: the guard / stream classes are its ...MasqueradingAsCUDA ones)
// and carries its backward as a C++ autograd node -- one dispatch per call where the ctypes mirror issues the
// library call plus 5 - 25 ATen ops from Python.  Built by hsg_amd/csrc/Makefile into libhsgk_torch.so (g++;
// links libhsgk.so and torch), loaded by hsg_amd/_torch_ops.py with torch.ops.load_library.
//
//   hsgk::segment_reduce(x, labels, P, mode) -> (out, status)
//       hsg/utils/segsort/common.py:11-41 calculate_prototypes_from_labels (mode 0), general/common.py:123-147
//       segment_mean (mode 1), raw sums (mode 2); status int32[1] != 0: a label outside [0, P).
//   hsgk::segsort_nll(emb, inst, proto, sems[], psems[], kappas[], modes[], pixel_groups?, proto_groups?) -> nll [L, n]
//       hsg/utils/segsort/loss.py:15-82 (:85-130 with class-mask labels), up to three label sets in one pass.
//   hsgk::segment_by_kmeans(x, labels?, loc, ...) -> (emb, emb_loc, labels, cluster, batch, meta_host, meta)
//       hsg/utils/segsort/common.py:270-408, one attempt of the library call; backward = hsgk_segment_by_kmeans_bwd.
//   hsgk::exchange_local(emb, emb_loc, cluster, batch, sem, inst, cap) ->
//       (prototypes, prototypes_with_loc, proto_sem, proto_inst, proto_batch, updated_cluster, meta_host)
//       hsg/models/utils.py:127-217 gather_clustering_and_update_prototypes for ONE rank / one device (no collective:
//       the multi-rank exchange keeps its two collectives between the phases, hsg_amd/models/utils.py).  meta_host:
//       int64[8] on the host = {rows of this rank, table rows, error bits, capacity needed, distinct images, most
//       segments of one image, ...}; with a capacity error the tensors are empty and the caller regrows `cap`.
//   hsgk::topk_prototypes (hsg/utils/segsort/eval.py:9-52) and hsgk::dmon_pool (hsg/utils/graph/loss.py:62-94, with its
//       one-launch backward) (round 6)
//   hsgk::pad_prototype_tables / hier_assign / group_mean / gather_labels (round 6)
//       hsg/models/embeddings/resnet_fcn_hsg.py:499-577, :638-672, :683-748, :751-780: the padded per-image tables,
//       the two-level assignment (+ its one-launch backward), the masked group means (+ hsgk_group_mean_bwd) and
//       the pixel -> group lookup -- one dispatch each, outputs allocated here.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <cstring>
#include <tuple>
#include <vector>

#include "../../include/hsgk.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

constexpr float kEps = 1e-12f;            // hsg/utils/general/common.py:101 (normalize_embedding's eps)
constexpr int64_t kChunk = 2048;          // HSGK_CHUNK (hsg_amd/_lib.py: CHUNK)
constexpr int64_t kErrCapacity = 2, kErrRows = 8;

void check(int rc, const char *what) {
  TORCH_CHECK(rc == 0, "libhsgk error ", rc, " in ", what, ": ", hsgk_last_error());
}

hsgk_stream_t stream_of(const Tensor &t) {
  return reinterpret_cast<hsgk_stream_t>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream());
}

Tensor rows_f32(const Tensor &t) {        // [.., d] -> contiguous float32 [n, d] (autograd-visible ops only where needed)
  Tensor r = t.dim() == 2 ? t : t.reshape({-1, t.size(-1)});
  if (r.scalar_type() != at::kFloat) r = r.to(at::kFloat);
  return r.contiguous();
}

Tensor vec_i64(const Tensor &t, const at::Device &dev) {
  Tensor r = t.dim() == 1 ? t : t.reshape({-1});
  if (r.scalar_type() != at::kLong || r.device() != dev) r = r.to(at::TensorOptions().dtype(at::kLong).device(dev));
  return r.contiguous();
}

// ---------------------------------------------------------------------------------------------------------
struct SegmentReduceFn : public torch::autograd::Function<SegmentReduceFn> {
  static variable_list forward(AutogradContext *ctx, const Tensor &x, const Tensor &labels, int64_t P, int64_t mode) {
    TORCH_CHECK(x.is_cuda(), "hsgk::segment_reduce: x must be on a ROCm device (there is no CPU path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t n = x.size(0), d = x.size(1);
    auto f32 = x.options().dtype(at::kFloat);
    Tensor out = at::empty({P, d}, f32), aux = at::empty({std::max<int64_t>(P, 1)}, f32);
    Tensor status = at::empty({1}, x.options().dtype(at::kInt));
    const size_t wsb = hsgk_segment_reduce_workspace_bytes(n, (int)d, P);
    Tensor ws = at::empty({(int64_t)wsb}, x.options().dtype(at::kByte));
    check(hsgk_segment_reduce(x.data_ptr<float>(), n, (int)d, labels.data_ptr<int64_t>(), P, (int)mode, kEps,
                              out.data_ptr<float>(), aux.data_ptr<float>(), status.data_ptr<int32_t>(), ws.data_ptr(),
                              wsb, stream_of(x)),
          "hsgk_segment_reduce");
    ctx->save_for_backward({out, aux, labels});
    ctx->saved_data["n"] = n;
    ctx->saved_data["mode"] = mode;
    ctx->mark_non_differentiable({status});
    return {out, status};
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &out = saved[0], &aux = saved[1], &labels = saved[2];
    const int64_t n = ctx->saved_data["n"].toInt(), mode = ctx->saved_data["mode"].toInt();
    const int64_t P = out.size(0), d = out.size(1);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(out.device());
    Tensor gout = grads[0].to(at::kFloat).contiguous();
    Tensor gseg = at::empty({std::max<int64_t>(P, 1), d}, out.options()), gx = at::empty({n, d}, out.options());
    check(hsgk_segment_reduce_bwd(gout.data_ptr<float>(), out.data_ptr<float>(), aux.data_ptr<float>(),
                                  labels.data_ptr<int64_t>(), n, (int)d, P, (int)mode, kEps, gseg.data_ptr<float>(),
                                  gx.data_ptr<float>(), stream_of(out)),
          "hsgk_segment_reduce_bwd");
    return {gx, Tensor(), Tensor(), Tensor()};
  }
};

std::tuple<Tensor, Tensor> segment_reduce(const Tensor &x, const Tensor &labels, int64_t P, int64_t mode) {
  Tensor xr = rows_f32(x);
  Tensor lab = vec_i64(labels, xr.device());
  TORCH_CHECK(lab.size(0) == xr.size(0), "hsgk::segment_reduce: one label per row");
  auto r = SegmentReduceFn::apply(xr, lab, P, mode);
  return {r[0], r[1]};
}

// ---------------------------------------------------------------------------------------------------------
// pinned host block + event per (thread, device) for the exchange's one host read
struct PinnedMeta {
  Tensor pin;
  hipEvent_t ev = nullptr;
};
PinnedMeta &pinned_meta(int dev) {
  // one heap-allocated pool per thread, never freed: its pinned blocks and events would otherwise be released by
  // thread_local destructors at thread / process exit, in an unspecified order against torch's host allocator and
  // the HIP runtime (64 bytes + one event per thread and device).  A block is used by one op at a time: the ops
  // wait for their read before they return, so a thread cannot re-enter with the block in flight.
  thread_local std::vector<PinnedMeta> *pool_p = new std::vector<PinnedMeta>();
  std::vector<PinnedMeta> &pool = *pool_p;
  if ((int)pool.size() <= dev) pool.resize(dev + 1);
  PinnedMeta &p = pool[dev];
  if (!p.pin.defined()) {
    p.pin = at::empty({8}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    TORCH_CHECK(hipEventCreateWithFlags(&p.ev, hipEventDisableTiming) == hipSuccess, "hipEventCreate");
  }
  return p;
}

struct ExchangeLocalFn : public torch::autograd::Function<ExchangeLocalFn> {
  static variable_list forward(AutogradContext *ctx, const Tensor &emb, const Tensor &emb_loc, const Tensor &c,
                               const Tensor &b, const Tensor &sem, const Tensor &inst, int64_t cap) {
    TORCH_CHECK(emb.is_cuda(), "hsgk::exchange_local: tensors must be on a ROCm device (there is no CPU path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(emb.device());
    const int64_t n = emb.size(0), C = emb.size(1), D = emb_loc.size(1);
    TORCH_CHECK(emb_loc.size(0) == n && c.size(0) == n && b.size(0) == n && sem.size(0) == n && inst.size(0) == n,
                "embeddings, embeddings_with_loc and the index vectors disagree on the number of pixels");
    auto f32 = emb.options().dtype(at::kFloat);
    auto i64 = emb.options().dtype(at::kLong);
    const int64_t nch = (n + kChunk - 1) / kChunk;
    const int64_t pool_rows = std::max<int64_t>(1, std::min<int64_t>(n, nch * std::min<int64_t>(cap, 256)));
    const size_t wsb = hsgk_exchange_workspace_bytes(n, (int)C, (int)D, cap, cap, 1, pool_rows);
    Tensor ws = at::empty({(int64_t)wsb}, emb.options().dtype(at::kByte));
    Tensor table = at::empty({cap, C + D}, f32), upd = at::empty({n}, i64), plab = at::empty({3, cap}, i64);
    Tensor meta = at::empty({8}, i64);
    hsgk_exchange_args a{};
    a.embeddings = emb.data_ptr<float>();
    a.embeddings_loc = emb_loc.data_ptr<float>();
    a.cluster = c.data_ptr<int64_t>();
    a.batch = b.data_ptr<int64_t>();
    a.semantic = sem.data_ptr<int64_t>();
    a.instance = inst.data_ptr<int64_t>();
    a.n = n; a.C = (int32_t)C; a.D = (int32_t)D;
    a.cap_local = cap; a.cap_total = cap; a.pool_rows = pool_rows; a.eps = kEps;
    a.table = table.data_ptr<float>();
    a.proto_semantic = plab.data_ptr<int64_t>();
    a.proto_instance = plab.data_ptr<int64_t>() + cap;
    a.proto_batch = plab.data_ptr<int64_t>() + 2 * cap;
    a.updated_cluster = upd.data_ptr<int64_t>();
    a.meta = meta.data_ptr<int64_t>();
    a.workspace = ws.data_ptr();
    a.workspace_bytes = wsb;
    hsgk_stream_t st = stream_of(emb);
    hipStream_t hst = reinterpret_cast<hipStream_t>(st);
    check(hsgk_exchange_keys(&a, 1, st), "hsgk_exchange_keys");
    check(hsgk_exchange_merge(&a, 0, 1, nullptr, st), "hsgk_exchange_merge");
    // the meta block is complete after the merge: it starts its way to the host now and the sums run behind the read
    PinnedMeta &pm = pinned_meta(emb.device().index());
    TORCH_CHECK(hipMemcpyAsync(pm.pin.data_ptr(), meta.data_ptr(), 8 * sizeof(int64_t), hipMemcpyDeviceToHost, hst) ==
                    hipSuccess, "hipMemcpyAsync(meta)");
    TORCH_CHECK(hipEventRecord(pm.ev, hst) == hipSuccess, "hipEventRecord");
    check(hsgk_exchange_sums(&a, 0, 1, nullptr, st), "hsgk_exchange_sums");
    TORCH_CHECK(hipEventSynchronize(pm.ev) == hipSuccess, "hipEventSynchronize");
    Tensor meta_host = at::empty({8}, at::TensorOptions().dtype(at::kLong));
    std::memcpy(meta_host.data_ptr(), pm.pin.data_ptr(), 8 * sizeof(int64_t));
    const int64_t *m = meta_host.data_ptr<int64_t>();
    const int64_t rows = m[1], err = m[2];
    Tensor pa, pb, norms, psem, pinst, pbatch;
    if (err) {                               // capacity: the caller regrows and repeats; anything else: it raises
      pa = at::empty({0, C}, f32); pb = at::empty({0, D}, f32); norms = at::empty({0, 2}, f32);
      psem = at::empty({0}, i64); pinst = at::empty({0}, i64); pbatch = at::empty({0}, i64);
    } else {
      pa = at::empty({rows, C}, f32); pb = at::empty({rows, D}, f32); norms = at::empty({rows, 2}, f32);
      if (rows) {
        hsgk_exchange_args f{};
        f.C = (int32_t)C; f.D = (int32_t)D; f.cap_total = rows; f.eps = kEps;
        f.table = table.data_ptr<float>();
        f.prototypes = pa.data_ptr<float>();
        f.prototypes_loc = pb.data_ptr<float>();
        f.norms = norms.data_ptr<float>();
        check(hsgk_exchange_finish(&f, rows, nullptr, 1, st), "hsgk_exchange_finish");
      }
      psem = plab.select(0, 0).narrow(0, 0, rows);
      pinst = plab.select(0, 1).narrow(0, 0, rows);
      pbatch = plab.select(0, 2).narrow(0, 0, rows);
    }
    ctx->save_for_backward({pa, pb, norms, upd});
    ctx->mark_non_differentiable({psem, pinst, pbatch, upd, meta_host});
    return {pa, pb, psem, pinst, pbatch, upd, meta_host};
  }

  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &pa = saved[0], &pb = saved[1], &norms = saved[2], &upd = saved[3];
    const int64_t rows = pa.size(0), C = pa.size(1), D = pb.size(1), n = upd.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(pa.device());
    hsgk_stream_t st = stream_of(pa);
    auto f32 = pa.options();
    // d(loss)/d(raw sums) of both normalised tables (mirror of hsgk_exchange_finish) ...
    Tensor gseg[2];
    const Tensor *outs[2] = {&pa, &pb};
    const int64_t width[2] = {C, D};
    for (int i = 0; i < 2; ++i) {
      if (!grads[i].defined() || rows == 0) continue;
      Tensor gp = grads[i].to(at::kFloat).contiguous();
      Tensor aux = norms.select(1, i).contiguous();
      gseg[i] = at::empty({rows, width[i]}, f32);
      check(hsgk_segment_reduce_bwd(gp.data_ptr<float>(), outs[i]->data_ptr<float>(), aux.data_ptr<float>(), nullptr, 0,
                                    (int)width[i], rows, 0, kEps, gseg[i].data_ptr<float>(), nullptr, st),
            "hsgk_segment_reduce_bwd(table)");
    }
    // ... and every pixel row receives the gradient of its segment's sum
    const bool need[2] = {ctx->needs_input_grad(0), ctx->needs_input_grad(1)};
    Tensor gx[2];
    for (int i = 0; i < 2; ++i) {
      if (!need[i]) continue;
      if (!gseg[i].defined()) { gx[i] = at::zeros({n, width[i]}, f32); continue; }
      gx[i] = at::empty({n, width[i]}, f32);
      check(hsgk_segment_reduce_bwd(gseg[i].data_ptr<float>(), gseg[i].data_ptr<float>(), nullptr, upd.data_ptr<int64_t>(), n,
                                    (int)width[i], rows, 2, kEps, gseg[i].data_ptr<float>(), gx[i].data_ptr<float>(), st),
            "hsgk_segment_reduce_bwd(rows)");
    }
    return {gx[0], gx[1], Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> exchange_local(
    const Tensor &emb, const Tensor &emb_loc, const Tensor &cluster, const Tensor &batch, const Tensor &sem,
    const Tensor &inst, int64_t cap) {
  Tensor e = rows_f32(emb), el = rows_f32(emb_loc);
  const at::Device dev = e.device();
  auto r = ExchangeLocalFn::apply(e, el, vec_i64(cluster, dev), vec_i64(batch, dev), vec_i64(sem, dev),
                                  vec_i64(inst, dev), cap);
  return {r[0], r[1], r[2], r[3], r[4], r[5], r[6]};
}

// ---------------------------------------------------------------------------------------------------------
// hsg/utils/segsort/loss.py:15-82 (and :85-130): per-pixel negative log-likelihood of up to three label sets over
// the same embeddings / own-prototype indices / prototype table in one pass over E P^T -> nll [L, n].
struct SegsortNllFn : public torch::autograd::Function<SegsortNllFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &emb, const Tensor &inst, const Tensor &proto,
                        std::vector<Tensor> sems, std::vector<Tensor> psems, std::vector<double> kappas,
                        std::vector<int64_t> modes, const c10::optional<Tensor> &qg, const c10::optional<Tensor> &pg) {
    TORCH_CHECK(emb.is_cuda(), "hsgk::segsort_nll: tensors must be on a ROCm device (there is no CPU path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(emb.device());
    const int64_t L = (int64_t)kappas.size(), n = emb.size(0), c = emb.size(1), P = proto.size(0);
    TORCH_CHECK(L >= 1 && L <= HSGK_LOSS_MAX_SETS && (int64_t)sems.size() == L && (int64_t)psems.size() == L &&
                    (int64_t)modes.size() == L, "1..3 label sets per pass");
    auto f32 = emb.options().dtype(at::kFloat);
    Tensor nll = at::empty({L, n}, f32), num = at::empty({L, n}, f32), den = at::empty({L, n}, f32);
    Tensor use_same = at::empty({L, n}, emb.options().dtype(at::kInt));
    const size_t wsb = hsgk_segsort_loss_workspace_bytes(n, (int)c, P, (int)L);
    Tensor ws = at::empty({(int64_t)wsb}, emb.options().dtype(at::kByte));
    hsgk_loss_set sets[HSGK_LOSS_MAX_SETS];
    for (int64_t l = 0; l < L; ++l)
      sets[l] = hsgk_loss_set{sems[l].data_ptr<int64_t>(), psems[l].data_ptr<int64_t>(), (float)kappas[l], (int32_t)modes[l]};
    const bool grouped = qg.has_value() && qg->defined();
    check(hsgk_segsort_loss_fwd(emb.data_ptr<float>(), n, (int)c, inst.data_ptr<int64_t>(), proto.data_ptr<float>(), P,
                                (int)L, sets, grouped ? qg->data_ptr<int64_t>() : nullptr,
                                grouped ? pg->data_ptr<int64_t>() : nullptr, nll.data_ptr<float>(), num.data_ptr<float>(),
                                den.data_ptr<float>(), use_same.data_ptr<int32_t>(), ws.data_ptr(), wsb, stream_of(emb)),
          "hsgk_segsort_loss_fwd");
    variable_list keep = {emb, proto, inst, num, den, use_same};
    for (auto &t : sems) keep.push_back(t);
    for (auto &t : psems) keep.push_back(t);
    if (grouped) { keep.push_back(*qg); keep.push_back(*pg); }
    ctx->save_for_backward(keep);
    ctx->saved_data["kappas"] = kappas;
    ctx->saved_data["modes"] = modes;
    ctx->saved_data["grouped"] = grouped;
    return nll;
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto sv = ctx->get_saved_variables();
    const Tensor &emb = sv[0], &proto = sv[1], &inst = sv[2], &num = sv[3], &den = sv[4], &use_same = sv[5];
    const auto kappas = ctx->saved_data["kappas"].toDoubleVector();
    const auto modes = ctx->saved_data["modes"].toIntVector();
    const bool grouped = ctx->saved_data["grouped"].toBool();
    const int64_t L = (int64_t)kappas.size(), n = emb.size(0), c = emb.size(1), P = proto.size(0);
    c10::hip::HIPGuardMasqueradingAsCUDA guard(emb.device());
    const bool want_e = ctx->needs_input_grad(0), want_p = ctx->needs_input_grad(2);
    Tensor g_emb, g_proto;
    if (want_e || want_p) {
      Tensor gscale = grads[0].to(at::kFloat).contiguous();
      if (want_e) g_emb = at::empty({n, c}, emb.options());
      if (want_p) g_proto = at::empty({P, c}, emb.options());
      const size_t wsb = hsgk_segsort_loss_bwd_workspace_bytes(n, (int)c, P, (int)L);
      Tensor ws = at::empty({(int64_t)wsb}, emb.options().dtype(at::kByte));
      hsgk_loss_set sets[HSGK_LOSS_MAX_SETS];
      for (int64_t l = 0; l < L; ++l)
        sets[l] = hsgk_loss_set{sv[6 + l].data_ptr<int64_t>(), sv[6 + L + l].data_ptr<int64_t>(), (float)kappas[l],
                                (int32_t)modes[l]};
      check(hsgk_segsort_loss_bwd(emb.data_ptr<float>(), n, (int)c, inst.data_ptr<int64_t>(), proto.data_ptr<float>(), P,
                                  (int)L, sets, grouped ? sv[6 + 2 * L].data_ptr<int64_t>() : nullptr,
                                  grouped ? sv[7 + 2 * L].data_ptr<int64_t>() : nullptr, num.data_ptr<float>(),
                                  den.data_ptr<float>(), use_same.data_ptr<int32_t>(), gscale.data_ptr<float>(),
                                  want_e ? g_emb.data_ptr<float>() : nullptr, want_p ? g_proto.data_ptr<float>() : nullptr,
                                  ws.data_ptr(), wsb, stream_of(emb)),
            "hsgk_segsort_loss_bwd");
    }
    return {g_emb, Tensor(), g_proto, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor segsort_nll(const Tensor &emb, const Tensor &inst, const Tensor &proto, at::TensorList sems, at::TensorList psems,
                   at::ArrayRef<double> kappas, at::IntArrayRef modes, const c10::optional<Tensor> &qg,
                   const c10::optional<Tensor> &pg) {
  Tensor e = rows_f32(emb), p = rows_f32(proto);
  const at::Device dev = e.device();
  std::vector<Tensor> a, b;
  for (size_t l = 0; l < sems.size(); ++l) {
    a.push_back(vec_i64(sems[l], dev));
    b.push_back(vec_i64(psems[l], dev));
    const int64_t words = std::max<int64_t>(modes[l] >> 8, 1);            // class-mask words per row (1: plain labels)
    TORCH_CHECK(a.back().size(0) == e.size(0) * words && b.back().size(0) == p.size(0) * words,
                "label vectors do not match the embeddings / prototypes");
  }
  c10::optional<Tensor> q, g;
  if (qg.has_value() && qg->defined()) {
    TORCH_CHECK(pg.has_value() && pg->defined(), "pixel and prototype groups come together");
    q = vec_i64(*qg, dev);
    g = vec_i64(*pg, dev);
    TORCH_CHECK(q->size(0) == e.size(0) && g->size(0) == p.size(0), "group vectors do not match the embeddings / prototypes");
  }
  return SegsortNllFn::apply(e, vec_i64(inst, dev), p, a, b, kappas.vec(), modes.vec(), q, g);
}

// ---------------------------------------------------------------------------------------------------------
// hsg/utils/segsort/common.py:270-408 segment_by_kmeans: ONE attempt of the library call (the rare repeats -- a
// label range the presence table cannot hold, a timed-out small-map wait -- are decided by the Python mirror from
// meta_host).  x [B,C,H,W] f32; labels [B,H,W] i64 or undefined; loc / seed_map as hsgk_segkm_args describes them.
// Returns the five outputs cut to the kept rows and meta_host int64[8] (n_rows, n_segments, label_min, label_max,
// n_chunks, error, ...).  Without a label map nothing is read from the device (the row count is B H W).
struct SegmentByKmeansFn : public torch::autograd::Function<SegmentByKmeansFn> {
  static variable_list forward(AutogradContext *ctx, const Tensor &x, const c10::optional<Tensor> &lab, const Tensor &loc,
                               int64_t loc_sb, const Tensor &seed_map, int64_t seed_sb, int64_t K, bool has_ignore,
                               int64_t ign, int64_t iterations, int64_t batch_offset, int64_t table_cap, int64_t flags,
                               bool want_grad) {
    TORCH_CHECK(x.is_cuda(), "hsgk::segment_by_kmeans: embeddings must be on a ROCm device (there is no CPU path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    const int64_t B = x.size(0), C = x.size(1), H = x.size(2), W = x.size(3), n_max = B * H * W;
    const bool labelled = lab.has_value() && lab->defined();
    auto f32 = x.options().dtype(at::kFloat);
    auto i64 = x.options().dtype(at::kLong);
    Tensor emb = at::empty({n_max, C}, f32), eloc = at::empty({n_max, C + 2}, f32);
    Tensor out_lab = at::empty({n_max}, i64), cluster = at::empty({n_max}, i64), batch = at::empty({n_max}, i64);
    Tensor meta = at::empty({8}, i64);
    Tensor norms, rowmap;
    if (want_grad) {
      norms = at::empty({n_max, 2}, f32);
      if (has_ignore) rowmap = at::empty({n_max}, i64);
    }
    const size_t wsb = hsgk_segment_by_kmeans_workspace_bytes((int)B, (int)C, (int)H, (int)W, (int)K, table_cap);
    Tensor ws = at::empty({(int64_t)wsb}, x.options().dtype(at::kByte));
    hsgk_segkm_args a{};
    a.embeddings = x.data_ptr<float>();
    a.labels = labelled ? lab->data_ptr<int64_t>() : nullptr;
    a.loc = loc.data_ptr<float>();
    a.loc_batch_stride = loc_sb;
    a.seed_map = seed_map.data_ptr<int32_t>();
    a.B = (int32_t)B; a.C = (int32_t)C; a.H = (int32_t)H; a.W = (int32_t)W; a.K = (int32_t)K;
    a.iterations = (int32_t)iterations;
    a.has_ignore = has_ignore ? 1 : 0;
    a.ignore_index = ign;
    a.batch_offset = batch_offset;
    a.table_cap = table_cap;
    a.out_embeddings = emb.data_ptr<float>();
    a.out_embeddings_loc = eloc.data_ptr<float>();
    a.out_labels = out_lab.data_ptr<int64_t>();
    a.out_cluster = cluster.data_ptr<int64_t>();
    a.out_batch = batch.data_ptr<int64_t>();
    a.meta = reinterpret_cast<hsgk_segkm_meta *>(meta.data_ptr<int64_t>());
    a.out_norms = norms.defined() ? norms.data_ptr<float>() : nullptr;
    a.out_rowmap = rowmap.defined() ? rowmap.data_ptr<int64_t>() : nullptr;
    a.workspace = ws.data_ptr();
    a.workspace_bytes = wsb;
    a.seed_batch_stride = seed_sb;
    a.flags = (int32_t)flags;
    hsgk_stream_t st = stream_of(x);
    check(hsgk_segment_by_kmeans(&a, st), "hsgk_segment_by_kmeans");
    Tensor meta_host = at::zeros({8}, at::TensorOptions().dtype(at::kLong));
    int64_t n = n_max;
    if (labelled) {                          // the operator's single host read
      hipStream_t hst = reinterpret_cast<hipStream_t>(st);
      PinnedMeta &pm = pinned_meta(x.device().index());
      TORCH_CHECK(hipMemcpyAsync(pm.pin.data_ptr(), meta.data_ptr(), 8 * sizeof(int64_t), hipMemcpyDeviceToHost, hst) ==
                      hipSuccess, "hipMemcpyAsync(meta)");
      TORCH_CHECK(hipEventRecord(pm.ev, hst) == hipSuccess, "hipEventRecord");
      TORCH_CHECK(hipEventSynchronize(pm.ev) == hipSuccess, "hipEventSynchronize");
      std::memcpy(meta_host.data_ptr(), pm.pin.data_ptr(), 8 * sizeof(int64_t));
      n = meta_host.data_ptr<int64_t>()[5] ? 0 : meta_host.data_ptr<int64_t>()[0];
    } else {
      meta_host.data_ptr<int64_t>()[0] = n_max;
    }
    Tensor e = emb.narrow(0, 0, n), el = eloc.narrow(0, 0, n);
    Tensor l = out_lab.narrow(0, 0, n), c = cluster.narrow(0, 0, n), bt = batch.narrow(0, 0, n);
    if (want_grad) {
      ctx->save_for_backward({e, el, norms, rowmap.defined() ? rowmap : Tensor()});
      ctx->saved_data["shape"] = std::vector<int64_t>{B, C, H, W};
    }
    // (the device meta block rides along for the deferred small-map time-out flag of the unlabelled call)
    ctx->mark_non_differentiable({l, c, bt, meta_host, meta});
    return {e, el, l, c, bt, meta_host, meta};
  }
  static variable_list backward(AutogradContext *ctx, variable_list grads) {
    auto sv = ctx->get_saved_variables();
    const Tensor &emb = sv[0], &eloc = sv[1], &norms = sv[2], &rowmap = sv[3];
    const auto shape = ctx->saved_data["shape"].toIntVector();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(emb.device());
    Tensor ge, gl;
    if (grads[0].defined()) ge = grads[0].to(at::kFloat).contiguous();
    if (grads[1].defined()) gl = grads[1].to(at::kFloat).contiguous();
    Tensor gx = at::empty({shape[0], shape[1], shape[2], shape[3]}, emb.options());
    check(hsgk_segment_by_kmeans_bwd(ge.defined() ? ge.data_ptr<float>() : nullptr, gl.defined() ? gl.data_ptr<float>() : nullptr,
                                     emb.data_ptr<float>(), eloc.data_ptr<float>(), norms.data_ptr<float>(),
                                     rowmap.defined() ? rowmap.data_ptr<int64_t>() : nullptr, (int)shape[0], (int)shape[1],
                                     (int)shape[2], (int)shape[3], kEps, gx.data_ptr<float>(), stream_of(emb)),
          "hsgk_segment_by_kmeans_bwd");
    variable_list out(14);
    out[0] = gx;
    return out;
  }
};

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> segment_by_kmeans(
    const Tensor &x, const c10::optional<Tensor> &labels, const Tensor &loc, int64_t loc_sb, const Tensor &seed_map,
    int64_t seed_sb, int64_t K, bool has_ignore, int64_t ign, int64_t iterations, int64_t batch_offset, int64_t table_cap,
    int64_t flags) {
  TORCH_CHECK(x.dim() == 4 && x.scalar_type() == at::kFloat, "embeddings must be float32 [B, C, H, W]");
  Tensor xc = x.contiguous();
  auto r = SegmentByKmeansFn::apply(xc, labels, loc, loc_sb, seed_map, seed_sb, K, has_ignore, ign, iterations,
                                    batch_offset, table_cap, flags, x.requires_grad() && at::GradMode::is_enabled());
  return {r[0], r[1], r[2], r[3], r[4], r[5], r[6]};
}

// ---------------------------------------------------------------------------------------------------------
// The hierarchy entry points (round 6): hsg/models/embeddings/resnet_fcn_hsg.py:455-780.
struct HierAssignFn : public torch::autograd::Function<HierAssignFn> {
  static variable_list forward(AutogradContext *ctx, const Tensor &fine_logits, const c10::optional<Tensor> &coarse_logits) {
    TORCH_CHECK(fine_logits.is_cuda(), "hsgk::hier_assign: tensors must be on a ROCm device (there is no CPU path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(fine_logits.device());
    Tensor fl = fine_logits.detach().to(at::kFloat).contiguous();
    const bool has_c = coarse_logits.has_value() && coarse_logits->defined();
    Tensor cl = has_c ? coarse_logits->detach().to(at::kFloat).contiguous() : Tensor();
    const int64_t B = fl.size(0), KF = fl.size(1), N = fl.size(2), KC = has_c ? cl.size(1) : 0;
    auto i64 = fl.options().dtype(at::kLong);
    Tensor fprob = at::empty_like(fl), flab = at::empty({B, N}, i64);
    Tensor cprob = at::empty({B, std::max<int64_t>(KC, 1), N}, fl.options()), clab = at::empty({B, N}, i64);
    check(hsgk_hier_assign(fl.data_ptr<float>(), has_c ? cl.data_ptr<float>() : nullptr, (int)B, (int)KF, (int)KC, (int)N,
                           fprob.data_ptr<float>(), flab.data_ptr<int64_t>(), has_c ? cprob.data_ptr<float>() : nullptr,
                           has_c ? clab.data_ptr<int64_t>() : nullptr, stream_of(fl)),
          "hsgk_hier_assign");
    ctx->save_for_backward({fl, has_c ? cl : at::empty({0}, fl.options())});
    ctx->saved_data["has_c"] = has_c;
    ctx->mark_non_differentiable({flab, clab});
    return {fprob, flab, cprob, clab};
  }
  static variable_list backward(AutogradContext *ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &fl = saved[0], &cl = saved[1];
    const bool has_c = ctx->saved_data["has_c"].toBool();
    c10::hip::HIPGuardMasqueradingAsCUDA guard(fl.device());
    const int64_t B = fl.size(0), KF = fl.size(1), N = fl.size(2), KC = has_c ? cl.size(1) : 0;
    Tensor g1 = g[0].defined() ? g[0].to(at::kFloat).contiguous() : Tensor();
    Tensor g2 = (has_c && g[2].defined()) ? g[2].to(at::kFloat).contiguous() : Tensor();
    Tensor gfl = at::empty_like(fl), gcl = has_c ? at::empty_like(cl) : Tensor();
    check(hsgk_hier_assign_bwd(fl.data_ptr<float>(), has_c ? cl.data_ptr<float>() : nullptr, (int)B, (int)KF, (int)KC,
                               (int)N, g1.defined() ? g1.data_ptr<float>() : nullptr,
                               g2.defined() ? g2.data_ptr<float>() : nullptr, gfl.data_ptr<float>(),
                               has_c ? gcl.data_ptr<float>() : nullptr, stream_of(fl)),
          "hsgk_hier_assign_bwd");
    return {gfl, gcl};
  }
};

std::tuple<Tensor, Tensor, Tensor, Tensor> hier_assign(const Tensor &fine_logits, const c10::optional<Tensor> &coarse_logits) {
  auto r = HierAssignFn::apply(fine_logits, coarse_logits);
  return {r[0], r[1], r[2], r[3]};
}

struct GroupMeanFn : public torch::autograd::Function<GroupMeanFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &prototypes, const Tensor &labels,
                        const c10::optional<Tensor> &masks, int64_t G, bool normalized) {
    TORCH_CHECK(prototypes.is_cuda(), "hsgk::group_mean: tensors must be on a ROCm device (there is no CPU path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(prototypes.device());
    Tensor p = prototypes.detach().to(at::kFloat).contiguous();
    Tensor lab = labels.to(at::kLong).contiguous();
    const bool has_m = masks.has_value() && masks->defined();
    Tensor mk = has_m ? masks->to(at::kByte).contiguous() : Tensor();
    const int64_t B = p.size(0), C = p.size(1), N = p.size(2);
    Tensor out = at::empty({B, C, G}, p.options());
    check(hsgk_group_mean(p.data_ptr<float>(), lab.data_ptr<int64_t>(), has_m ? mk.data_ptr<uint8_t>() : nullptr, (int)B,
                          (int)C, (int)N, (int)G, normalized ? 1 : 0, kEps, out.data_ptr<float>(), stream_of(p)),
          "hsgk_group_mean");
    ctx->save_for_backward({p, lab, has_m ? mk : at::empty({0}, p.options().dtype(at::kByte))});
    ctx->saved_data["G"] = G;
    ctx->saved_data["normalized"] = normalized;
    ctx->saved_data["has_m"] = has_m;
    return out;
  }
  static variable_list backward(AutogradContext *ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &p = saved[0], &lab = saved[1], &mk = saved[2];
    c10::hip::HIPGuardMasqueradingAsCUDA guard(p.device());
    const int64_t B = p.size(0), C = p.size(1), N = p.size(2), G = ctx->saved_data["G"].toInt();
    Tensor go = g[0].to(at::kFloat).contiguous(), gp = at::empty_like(p);
    check(hsgk_group_mean_bwd(p.data_ptr<float>(), lab.data_ptr<int64_t>(),
                              ctx->saved_data["has_m"].toBool() ? mk.data_ptr<uint8_t>() : nullptr, (int)B, (int)C, (int)N,
                              (int)G, ctx->saved_data["normalized"].toBool() ? 1 : 0, kEps, go.data_ptr<float>(),
                              gp.data_ptr<float>(), stream_of(p)),
          "hsgk_group_mean_bwd");
    return {gp, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor group_mean(const Tensor &prototypes, const Tensor &labels, const c10::optional<Tensor> &masks, int64_t G,
                  bool normalized) {
  return GroupMeanFn::apply(prototypes, labels, masks, G, normalized);
}

Tensor gather_labels(const Tensor &table, const Tensor &img, const Tensor &seg) {
  TORCH_CHECK(seg.is_cuda(), "hsgk::gather_labels: tensors must be on a ROCm device (there is no CPU path)");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(seg.device());
  Tensor t = table.to(at::kLong).contiguous(), s = vec_i64(seg, seg.device()), im = vec_i64(img, seg.device());
  Tensor out = at::empty_like(s);
  if (s.numel())
    check(hsgk_gather_labels(t.data_ptr<int64_t>(), (int)t.size(1), im.data_ptr<int64_t>(), s.data_ptr<int64_t>(), s.numel(),
                             out.data_ptr<int64_t>(), stream_of(s)),
          "hsgk_gather_labels");
  return out;
}

// rows[s] sit at table[slot[s]]: the gradient of a row is its table row's (the padded tables' autograd edge)
struct PlacedRowsFn : public torch::autograd::Function<PlacedRowsFn> {
  static Tensor forward(AutogradContext *ctx, const Tensor &rows, const Tensor &slot, const Tensor &table) {
    ctx->save_for_backward({slot});
    return table.view_as(table);
  }
  static variable_list backward(AutogradContext *ctx, variable_list g) {
    return {g[0].index_select(0, ctx->get_saved_variables()[0]), Tensor(), Tensor()};
  }
};

// (table [B*M, C], pos table [B*M, Cp] or empty, masks bool [B*M], labels / batch ids int64 [B*M], by_image int64 [n],
//  pixel_image int64 [n]); protos / pos differentiable through their table rows
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor> pad_prototype_tables(
    const Tensor &seg_image, const Tensor &protos, const c10::optional<Tensor> &pos, const Tensor &seg_lab,
    const Tensor &seg_batch, const Tensor &pixel_seg, int64_t B, int64_t M) {
  TORCH_CHECK(protos.is_cuda(), "hsgk::pad_prototype_tables: tensors must be on a ROCm device (there is no CPU path)");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(protos.device());
  const auto dev = protos.device();
  const bool has_p = pos.has_value() && pos->defined();
  Tensor pr = protos.detach().to(at::kFloat).contiguous(), po = has_p ? pos->detach().to(at::kFloat).contiguous() : Tensor();
  Tensor si = vec_i64(seg_image, dev), sl = vec_i64(seg_lab, dev), sb = vec_i64(seg_batch, dev), ps = vec_i64(pixel_seg, dev);
  const int64_t P = pr.size(0), C = pr.size(1), Cp = has_p ? po.size(1) : 0, n = ps.size(0);
  auto f32 = pr.options();
  auto i64 = pr.options().dtype(at::kLong);
  Tensor table = at::empty({B * M, C}, f32), ptab = has_p ? at::empty({B * M, Cp}, f32) : at::empty({0}, f32);
  Tensor masks = at::empty({B * M}, pr.options().dtype(at::kBool)), plabs = at::empty({B * M}, i64), pbatch = at::empty({B * M}, i64);
  Tensor by_image = at::empty({n}, i64), pixel_image = at::empty({n}, i64);
  Tensor work = at::empty({2 * P + B + 1}, pr.options().dtype(at::kInt));
  const bool need_grad = protos.requires_grad() || (has_p && pos->requires_grad());
  Tensor slot = need_grad ? at::empty({P}, i64) : Tensor();
  check(hsgk_pad_prototype_tables(si.data_ptr<int64_t>(), P, pr.data_ptr<float>(), (int)C, has_p ? po.data_ptr<float>() : nullptr,
                                  (int)Cp, sl.data_ptr<int64_t>(), sb.data_ptr<int64_t>(), ps.data_ptr<int64_t>(), n, (int)B,
                                  (int)M, table.data_ptr<float>(), has_p ? ptab.data_ptr<float>() : nullptr,
                                  reinterpret_cast<uint8_t *>(masks.data_ptr<bool>()), plabs.data_ptr<int64_t>(),
                                  pbatch.data_ptr<int64_t>(), by_image.data_ptr<int64_t>(), pixel_image.data_ptr<int64_t>(),
                                  need_grad ? slot.data_ptr<int64_t>() : nullptr, work.data_ptr<int32_t>(), stream_of(pr)),
        "hsgk_pad_prototype_tables");
  if (protos.requires_grad()) table = PlacedRowsFn::apply(protos, slot, table);
  if (has_p && pos->requires_grad()) ptab = PlacedRowsFn::apply(*pos, slot, ptab);
  return {table, ptab, masks, plabs, pbatch, by_image, pixel_image};
}

// ---------------------------------------------------------------------------------------------------------
// hsg/utils/segsort/eval.py:9-52 (+ hsg/models/utils.py:243-309 with groups): the k best prototypes per query
std::tuple<Tensor, Tensor> topk_prototypes(const Tensor &queries, const Tensor &prototypes, int64_t top_k,
                                           const c10::optional<Tensor> &query_groups,
                                           const c10::optional<Tensor> &prototype_groups) {
  TORCH_CHECK(queries.is_cuda(), "hsgk::topk_prototypes: tensors must be on a ROCm device (there is no CPU path)");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(queries.device());
  Tensor q = rows_f32(queries.detach()), p = rows_f32(prototypes.detach());
  const int64_t n = q.size(0), c = q.size(1), P = p.size(0);
  const bool grouped = query_groups.has_value() && query_groups->defined();
  Tensor qg = grouped ? vec_i64(*query_groups, q.device()) : Tensor();
  Tensor pg = grouped ? vec_i64(*prototype_groups, q.device()) : Tensor();
  Tensor idx = at::empty({n, top_k}, q.options().dtype(at::kLong)), val = at::empty({n, top_k}, q.options());
  const size_t wsb = hsgk_topk_workspace_bytes(n, (int)c, P, (int)top_k);
  Tensor ws = at::empty({(int64_t)wsb}, q.options().dtype(at::kByte));
  check(hsgk_topk_prototypes_grouped(q.data_ptr<float>(), n, (int)c, p.data_ptr<float>(), P, (int)top_k,
                                     grouped ? qg.data_ptr<int64_t>() : nullptr, grouped ? pg.data_ptr<int64_t>() : nullptr,
                                     idx.data_ptr<int64_t>(), val.data_ptr<float>(), ws.data_ptr(), wsb, stream_of(q)),
        "hsgk_topk_prototypes_grouped");
  return {idx, val};
}

// hsg/utils/graph/loss.py:62-94 for an adjacency without gradient: per image t = (Tr(S^T A S) - |S^T d|^2 / 2m) / 2m
// and c = |sum_i S_i|_2, one launch forward and one backward
struct DmonPoolFn : public torch::autograd::Function<DmonPoolFn> {
  static variable_list forward(AutogradContext *ctx, const Tensor &adj, const Tensor &s, const c10::optional<Tensor> &valid) {
    TORCH_CHECK(s.is_cuda(), "hsgk::dmon_pool: tensors must be on a ROCm device (there is no CPU path)");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(s.device());
    Tensor a = adj.detach().to(at::kFloat).contiguous(), sc = s.detach().to(at::kFloat).contiguous();
    const bool has_v = valid.has_value() && valid->defined();
    Tensor v = has_v ? valid->to(at::kByte).contiguous() : Tensor();
    const int64_t B = sc.size(0), N = sc.size(1), K = sc.size(2);
    Tensor t = at::empty({B}, sc.options()), c = at::empty({B}, sc.options());
    const size_t nb = hsgk_dmon_pool_workspace_bytes((int)B, (int)N, (int)K);
    Tensor saved = at::empty({(int64_t)nb}, sc.options().dtype(at::kByte));
    check(hsgk_dmon_pool_fwd(a.data_ptr<float>(), sc.data_ptr<float>(), has_v ? v.data_ptr<uint8_t>() : nullptr, (int)B, (int)N,
                             (int)K, t.data_ptr<float>(), c.data_ptr<float>(), saved.data_ptr(), nb, stream_of(sc)),
          "hsgk_dmon_pool_fwd");
    ctx->save_for_backward({a, sc, has_v ? v : at::empty({0}, sc.options().dtype(at::kByte)), saved});
    ctx->saved_data["has_v"] = has_v;
    return {t, c};
  }
  static variable_list backward(AutogradContext *ctx, variable_list g) {
    auto sv = ctx->get_saved_variables();
    const Tensor &a = sv[0], &sc = sv[1], &v = sv[2], &saved = sv[3];
    c10::hip::HIPGuardMasqueradingAsCUDA guard(sc.device());
    const int64_t B = sc.size(0), N = sc.size(1), K = sc.size(2);
    Tensor gt = g[0].defined() ? g[0].to(at::kFloat).contiguous() : at::zeros({B}, sc.options());
    Tensor gc = g[1].defined() ? g[1].to(at::kFloat).contiguous() : at::zeros({B}, sc.options());
    Tensor grad = at::empty_like(sc);
    check(hsgk_dmon_pool_bwd(a.data_ptr<float>(), sc.data_ptr<float>(),
                             ctx->saved_data["has_v"].toBool() ? v.data_ptr<uint8_t>() : nullptr, (int)B, (int)N, (int)K,
                             saved.data_ptr(), gt.data_ptr<float>(), gc.data_ptr<float>(), grad.data_ptr<float>(),
                             stream_of(sc)),
          "hsgk_dmon_pool_bwd");
    return {Tensor(), grad, Tensor()};
  }
};

std::tuple<Tensor, Tensor> dmon_pool(const Tensor &adj, const Tensor &s, const c10::optional<Tensor> &valid) {
  auto r = DmonPoolFn::apply(adj, s, valid);
  return {r[0], r[1]};
}

int64_t abi_version() { return hsgk_version(); }

}  // namespace

TORCH_LIBRARY(hsgk, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("segment_reduce(Tensor x, Tensor labels, int P, int mode) -> (Tensor, Tensor)", &segment_reduce);
  m.def("exchange_local(Tensor emb, Tensor emb_loc, Tensor cluster, Tensor batch, Tensor sem, Tensor inst, int cap)"
        " -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", &exchange_local);
  m.def("segsort_nll(Tensor emb, Tensor inst, Tensor proto, Tensor[] sems, Tensor[] psems, float[] kappas, int[] modes,"
        " Tensor? pixel_groups, Tensor? proto_groups) -> Tensor", &segsort_nll);
  m.def("topk_prototypes(Tensor queries, Tensor prototypes, int top_k, Tensor? query_groups, Tensor? prototype_groups)"
        " -> (Tensor, Tensor)", &topk_prototypes);
  m.def("dmon_pool(Tensor adj, Tensor s, Tensor? valid) -> (Tensor, Tensor)", &dmon_pool);
  m.def("hier_assign(Tensor fine_logits, Tensor? coarse_logits) -> (Tensor, Tensor, Tensor, Tensor)", &hier_assign);
  m.def("group_mean(Tensor prototypes, Tensor labels, Tensor? masks, int num_groups, bool normalized) -> Tensor", &group_mean);
  m.def("gather_labels(Tensor table, Tensor img, Tensor seg) -> Tensor", &gather_labels);
  m.def("pad_prototype_tables(Tensor seg_image, Tensor protos, Tensor? pos, Tensor seg_lab, Tensor seg_batch,"
        " Tensor pixel_seg, int B, int M) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", &pad_prototype_tables);
  m.def("segment_by_kmeans(Tensor x, Tensor? labels, Tensor loc, int loc_batch_stride, Tensor seed_map,"
        " int seed_batch_stride, int K, bool has_ignore, int ignore_index, int iterations, int batch_offset,"
        " int table_cap, int flags) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", &segment_by_kmeans);
}
