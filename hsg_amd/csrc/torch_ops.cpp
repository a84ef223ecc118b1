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
//   hsgk::exchange_local(emb, emb_loc, cluster, batch, sem, inst, cap) ->
//       (prototypes, prototypes_with_loc, proto_sem, proto_inst, proto_batch, updated_cluster, meta_host)
//       hsg/models/utils.py:127-217 gather_clustering_and_update_prototypes for ONE rank / one device (no collective:
//       the multi-rank exchange keeps its two collectives between the phases, hsg_amd/models/utils.py).  meta_host:
//       int64[8] on the host = {rows of this rank, table rows, error bits, capacity needed, distinct images, most
//       segments of one image, ...}; with a capacity error the tensors are empty and the caller regrows `cap`.
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
  thread_local std::vector<PinnedMeta> pool;
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

int64_t abi_version() { return hsgk_version(); }

}  // namespace

TORCH_LIBRARY(hsgk, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("segment_reduce(Tensor x, Tensor labels, int P, int mode) -> (Tensor, Tensor)", &segment_reduce);
  m.def("exchange_local(Tensor emb, Tensor emb_loc, Tensor cluster, Tensor batch, Tensor sem, Tensor inst, int cap)"
        " -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)", &exchange_local);
}
