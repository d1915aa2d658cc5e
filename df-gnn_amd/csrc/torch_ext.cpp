// torch_ext.cpp -- the thin torch/extension.h shim over the C ABI of libdfgnn.so (include/dfgnn.h).
//
// This is the binding a maintainer of the reference would keep: pybind11 + torch::Tensor like
// DFGNN/src/fused_gtconv/fused_gtconv.cpp:577-602 and DFGNN/src/fused_gatconv/fused_gatconv.cpp:355-372, with each
// *_cuda host launcher (fused_gtconv_hyper.cu:679-760, fused_gtconv_backward.cu:231-265, fused_gatconv_*.cu) replaced by
// a few lines over the C ABI.  It holds no kernels: argument checks (real ones -- the reference's dtype / shape asserts
// are compiled out), output allocation, device guard + torch's CURRENT stream, one call into libdfgnn.so.
// Built by dfgnn_native.build() into df-gnn_amd/_dfgnn_ext.so (g++, in-tree, no JIT cache); the Python modules
// fused_gtconv / fused_gatconv route their hot entry points through it (a ctypes call costs ~25-60 us of host time per
// operator, this ~5) and keep the ctypes path for everything else.  Both paths end in the same dfgnn_* symbols.
//
// The optional block plan (dfgnn_plan_build, cached per batch structure by the Python side) and the "edge values are all
// ones" flag are passed in as plain integers / bool: caching lives in _binding_util.py for both bindings.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>   // torch-ROCm: HIP devices are called "cuda"; these are the
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>      // guard / stream types behind torch.cuda.*
#include <torch/extension.h>

#include <vector>

#include "../../include/dfgnn.h"

#ifndef DFGNN_SRC_HASH
#define DFGNN_SRC_HASH "unknown"
#endif

namespace {

using torch::Tensor;

inline void check_cuda_contig(const Tensor &t, const char *name) {
  TORCH_CHECK(t.is_cuda(), name, " must be on CUDA");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
}
inline void check_i32(const Tensor &t, const char *name) {
  check_cuda_contig(t, name);
  TORCH_CHECK(t.scalar_type() == torch::kInt32, name, " must have dtype torch.int32, got ", t.scalar_type());
}
inline void check_f32(const Tensor &t, const char *name) {
  check_cuda_contig(t, name);
  TORCH_CHECK(t.scalar_type() == torch::kFloat32, name, " must have dtype torch.float32, got ", t.scalar_type());
}
inline void check_feat3(const Tensor &t, const Tensor &like, const char *name) {
  check_f32(t, name);
  TORCH_CHECK(t.dim() == 3, name, " must have shape [nodes, heads, feat], got ", t.sizes());
  TORCH_CHECK(t.sizes() == like.sizes(), name, " has shape ", t.sizes(), ", expected ", like.sizes());
}
inline void check_edges(const Tensor &t, int64_t nnz, const char *name) {
  TORCH_CHECK(t.dim() == 1 && t.size(0) == nnz, name, " must have shape (", nnz, ",), got ", t.sizes());
}
inline void check_same_device(const Tensor &ref, std::initializer_list<const Tensor *> ts) {
  for (const Tensor *t : ts)
    TORCH_CHECK(!t->defined() || t->device() == ref.device(), "every tensor must live on one device (", ref.device(), "), got ",
                t->device(), ": a pointer of another GPU would be handed to a kernel of this one");
}
inline void check_rc(int rc, const char *what) {
  TORCH_CHECK(rc == 0, what, " failed: ", dfgnn_error_string(rc), " (code ", rc, ")");
}
inline dfgnn_stream_t cur_stream() {  // torch's current stream of the (guarded) current device
  return reinterpret_cast<dfgnn_stream_t>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream());
}
inline const int *plan_ptr(int64_t p) { return reinterpret_cast<const int *>(static_cast<intptr_t>(p)); }

struct GtDims {
  int m, nnz, h, f;
};
GtDims gt_checks(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &rows, const Tensor &val, const Tensor &Q,
                 const Tensor &K, const Tensor &V) {
  check_i32(row_ptr, "row_ptr");
  check_i32(col_ind, "col_ind");
  check_i32(rows, "rows");
  check_f32(val, "val");
  check_feat3(Q, Q, "Q");
  check_feat3(K, Q, "K");
  check_feat3(V, Q, "V");
  TORCH_CHECK(row_ptr.dim() == 1 && col_ind.dim() == 1, "indptr / indices must be 1-D");
  TORCH_CHECK(row_ptr.size(0) - 1 == Q.size(0), "indptr describes ", row_ptr.size(0) - 1, " rows but features have ", Q.size(0),
              " nodes");
  const int64_t nnz = col_ind.size(0);
  check_edges(rows, nnz, "rows");
  check_edges(val, nnz, "val");
  check_same_device(Q, {&row_ptr, &col_ind, &rows, &val, &K, &V});
  return GtDims{(int)Q.size(0), (int)nnz, (int)Q.size(1), (int)Q.size(2)};
}

// fused_gtconv.cpp:278-314 (want_attn = false) and :79-116 (want_attn = true)
std::vector<Tensor> gt_hyper_fwd(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &rows, const Tensor &val,
                                 const Tensor &Q, const Tensor &K, const Tensor &V, bool want_attn, bool unit_val,
                                 int64_t plan, int64_t meta) {
  const GtDims d = gt_checks(row_ptr, col_ind, rows, val, Q, K, V);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor out = torch::empty_like(Q);
  Tensor attn, ws;
  if (want_attn) attn = torch::empty({d.h, d.nnz}, Q.options());
  else if (plan) ws = torch::empty({d.h, d.nnz}, Q.options());  // per-edge scratch of the inference call (include/dfgnn.h)
  check_rc(dfgnn_gt_hyper_fwd(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), rows.data_ptr<int>(),
                              unit_val ? nullptr : val.data_ptr<float>(), Q.data_ptr<float>(), K.data_ptr<float>(),
                              V.data_ptr<float>(), want_attn ? attn.data_ptr<float>() : nullptr,
                              ws.defined() ? ws.data_ptr<float>() : nullptr, out.data_ptr<float>(), plan_ptr(plan),
                              plan_ptr(meta), cur_stream()),
           want_attn ? "gt_hyper_forward" : "gt_hyper_inference");
  if (want_attn) return {out, attn};
  return {out};
}

// fused_gtconv.cpp:125-172
std::vector<Tensor> gt_bwd(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &rows, const Tensor &val,
                           const Tensor &col_ptr, const Tensor &row_ind, const Tensor &val_idx, const Tensor &Q,
                           const Tensor &K, const Tensor &V, const Tensor &attn_edge, const Tensor &grad, bool unit_val,
                           int64_t plan, int64_t meta) {
  const GtDims d = gt_checks(row_ptr, col_ind, rows, val, Q, K, V);
  check_i32(col_ptr, "col_ptr");
  check_i32(row_ind, "row_ind");
  check_i32(val_idx, "val_idx");
  check_f32(attn_edge, "attn_edge");
  check_feat3(grad, Q, "grad");
  check_edges(row_ind, d.nnz, "row_ind");
  check_edges(val_idx, d.nnz, "val_idx");
  TORCH_CHECK(col_ptr.dim() == 1 && col_ptr.size(0) == d.m + 1, "col_ptr must have shape (", d.m + 1,
              ",): the adjacency must be square");
  TORCH_CHECK(attn_edge.numel() == (int64_t)d.h * d.nnz, "attn_edge must have ", d.h, "*", d.nnz, " elements, got ",
              attn_edge.numel());
  check_same_device(Q, {&col_ptr, &row_ind, &val_idx, &attn_edge, &grad});
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor grad_edge = torch::empty({d.h, d.nnz}, Q.options());
  Tensor dQ = torch::empty_like(Q), dK = torch::empty_like(K), dV = torch::empty_like(V);
  check_rc(dfgnn_gt_bwd(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), rows.data_ptr<int>(),
                        unit_val ? nullptr : val.data_ptr<float>(), col_ptr.data_ptr<int>(), row_ind.data_ptr<int>(),
                        val_idx.data_ptr<int>(), Q.data_ptr<float>(), K.data_ptr<float>(), V.data_ptr<float>(),
                        attn_edge.data_ptr<float>(), grad.data_ptr<float>(), grad_edge.data_ptr<float>(),
                        dQ.data_ptr<float>(), dK.data_ptr<float>(), dV.data_ptr<float>(), plan_ptr(plan), plan_ptr(meta),
                        cur_stream()),
           "gt_backward");
  return {dQ, dK, dV};
}

// ---- the statistics-saving training pair (include/dfgnn.h: dfgnn_gt_hyper_fwd_stats / dfgnn_gt_bwd_stats) ---------------
GtDims gt_stats_checks(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &Q, const Tensor &K, const Tensor &V) {
  check_i32(row_ptr, "row_ptr");
  check_i32(col_ind, "col_ind");
  check_feat3(Q, Q, "Q");
  check_feat3(K, Q, "K");
  check_feat3(V, Q, "V");
  TORCH_CHECK(row_ptr.dim() == 1 && col_ind.dim() == 1, "indptr / indices must be 1-D");
  TORCH_CHECK(row_ptr.size(0) - 1 == Q.size(0), "indptr describes ", row_ptr.size(0) - 1, " rows but features have ", Q.size(0),
              " nodes");
  for (const Tensor *t : {&row_ptr, &col_ind, &K, &V})
    TORCH_CHECK(t->device() == Q.device(), "every tensor must live on the device of Q (", Q.device(), "), got ", t->device());
  return GtDims{(int)Q.size(0), (int)col_ind.size(0), (int)Q.size(1), (int)Q.size(2)};
}

// weights: the plan's dense edge values (plan_dense_weights below), or nothing for unit values
const float *weights_ptr(const c10::optional<Tensor> &weights, const Tensor &Q, int m) {
  if (!weights.has_value()) return nullptr;
  const Tensor &w = *weights;
  check_f32(w, "weights");
  TORCH_CHECK(w.numel() == (int64_t)dfgnn_plan_dense_weights_floats(m), "weights must hold ", dfgnn_plan_dense_weights_floats(m),
              " floats (dfgnn_plan_dense_weights), got ", w.numel());
  TORCH_CHECK(w.device() == Q.device(), "weights must live on the device of Q");
  return w.data_ptr<float>();
}

// save_stats = false: inference (nothing but `out` is produced; edge values on the matrix cores)
std::vector<Tensor> gt_hyper_fwd_stats(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &Q, const Tensor &K,
                                       const Tensor &V, int64_t plan, int64_t meta, const c10::optional<Tensor> &weights,
                                       bool save_stats) {
  const GtDims d = gt_stats_checks(row_ptr, col_ind, Q, K, V);
  const float *w = weights_ptr(weights, Q, d.m);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor out = torch::empty_like(Q);
  Tensor row_max, row_sum;
  if (save_stats) {
    row_max = torch::empty({d.m, d.h}, Q.options());
    row_sum = torch::empty({d.m, d.h}, Q.options());
  }
  check_rc(dfgnn_gt_hyper_fwd_stats(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), w, Q.data_ptr<float>(),
                                    K.data_ptr<float>(), V.data_ptr<float>(), save_stats ? row_max.data_ptr<float>() : nullptr,
                                    save_stats ? row_sum.data_ptr<float>() : nullptr, out.data_ptr<float>(), plan_ptr(plan),
                                    plan_ptr(meta), cur_stream()),
           "gt_hyper_forward_stats");
  if (!save_stats) return {out};
  return {out, row_max, row_sum};
}

std::vector<Tensor> gt_bwd_stats(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &Q, const Tensor &K,
                                 const Tensor &V, const Tensor &row_max, const Tensor &row_sum, const Tensor &grad,
                                 int64_t plan, int64_t meta, const c10::optional<Tensor> &weights) {
  const GtDims d = gt_stats_checks(row_ptr, col_ind, Q, K, V);
  const float *w = weights_ptr(weights, Q, d.m);
  check_feat3(grad, Q, "grad");
  check_f32(row_max, "row_max");
  check_f32(row_sum, "row_sum");
  for (const Tensor *t : {&row_max, &row_sum}) {
    TORCH_CHECK(t->dim() == 2 && t->size(0) == d.m && t->size(1) == d.h, "row_max / row_sum must have shape (", d.m, ", ", d.h,
                "), got ", t->sizes());
    TORCH_CHECK(t->device() == Q.device(), "row statistics must live on the device of Q");
  }
  TORCH_CHECK(grad.device() == Q.device(), "grad must live on the device of Q");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor dQ = torch::empty_like(Q), dK = torch::empty_like(K), dV = torch::empty_like(V);
  check_rc(dfgnn_gt_bwd_stats(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), w, Q.data_ptr<float>(),
                              K.data_ptr<float>(), V.data_ptr<float>(), row_max.data_ptr<float>(), row_sum.data_ptr<float>(),
                              grad.data_ptr<float>(), dQ.data_ptr<float>(), dK.data_ptr<float>(), dV.data_ptr<float>(),
                              plan_ptr(plan), plan_ptr(meta), cur_stream()),
           "gt_backward_stats");
  return {dQ, dK, dV};
}

// ---- the attn_edge pair in rank order (include/dfgnn.h: dfgnn_gt_hyper_fwd_ranked / dfgnn_gt_bwd_ranked) ----------------
std::vector<Tensor> gt_hyper_fwd_ranked(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &Q, const Tensor &K,
                                        const Tensor &V, int64_t plan, int64_t meta) {
  const GtDims d = gt_stats_checks(row_ptr, col_ind, Q, K, V);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor out = torch::empty_like(Q);
  Tensor attn = torch::empty({d.h, d.nnz}, Q.options());
  check_rc(dfgnn_gt_hyper_fwd_ranked(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), Q.data_ptr<float>(),
                                     K.data_ptr<float>(), V.data_ptr<float>(), attn.data_ptr<float>(), out.data_ptr<float>(),
                                     plan_ptr(plan), plan_ptr(meta), cur_stream()),
           "gt_hyper_forward_ranked");
  return {out, attn};
}

std::vector<Tensor> gt_bwd_ranked(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &Q, const Tensor &K,
                                  const Tensor &V, const Tensor &attn, const Tensor &grad, int64_t plan, int64_t meta) {
  const GtDims d = gt_stats_checks(row_ptr, col_ind, Q, K, V);
  check_feat3(grad, Q, "grad");
  check_f32(attn, "attn_ranked");
  TORCH_CHECK(attn.numel() == (int64_t)d.h * d.nnz, "attn_ranked must have ", d.h, "*", d.nnz, " elements, got ", attn.numel());
  TORCH_CHECK(attn.device() == Q.device() && grad.device() == Q.device(), "attn_ranked / grad must live on the device of Q");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor dQ = torch::empty_like(Q), dK = torch::empty_like(K), dV = torch::empty_like(V);
  check_rc(dfgnn_gt_bwd_ranked(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), Q.data_ptr<float>(),
                               K.data_ptr<float>(), V.data_ptr<float>(), attn.data_ptr<float>(), grad.data_ptr<float>(),
                               dQ.data_ptr<float>(), dK.data_ptr<float>(), dV.data_ptr<float>(), plan_ptr(plan), plan_ptr(meta),
                               cur_stream()),
           "gt_backward_ranked");
  return {dQ, dK, dV};
}

// the edge values of a plan's dense ranges in dense form (dfgnn_plan_dense_weights): fp32[256 m]
Tensor plan_dense_weights(const Tensor &row_ptr, const Tensor &val, int64_t plan, int64_t meta) {
  check_i32(row_ptr, "row_ptr");
  check_f32(val, "val");
  TORCH_CHECK(row_ptr.dim() == 1 && row_ptr.size(0) >= 1, "indptr must be 1-D");
  TORCH_CHECK(val.device() == row_ptr.device(), "val must live on the device of indptr");
  const int m = (int)row_ptr.size(0) - 1, nnz = (int)val.numel();
  c10::hip::HIPGuardMasqueradingAsCUDA guard(row_ptr.device());
  Tensor w = torch::empty({(int64_t)dfgnn_plan_dense_weights_floats(m)}, val.options());
  check_rc(dfgnn_plan_dense_weights(m, nnz, row_ptr.data_ptr<int>(), val.data_ptr<float>(), plan_ptr(plan), plan_ptr(meta),
                                    w.data_ptr<float>(), cur_stream()),
           "plan_dense_weights");
  return w;
}

struct GatDims {
  int m, nnz, h, f;
};
GatDims gat_checks(const Tensor &attn_row, const Tensor &attn_col, const Tensor &indptr, const Tensor &indices,
                   const Tensor *rows, const Tensor &in_feat) {
  check_f32(attn_row, "attn_row");
  check_f32(attn_col, "attn_col");
  check_f32(in_feat, "in_feat");
  check_i32(indptr, "indptr");
  check_i32(indices, "indices");
  TORCH_CHECK(in_feat.dim() == 3, "in_feat must have shape [nodes, heads, feat], got ", in_feat.sizes());
  const int64_t m = indptr.size(0) - 1, nnz = indices.size(0);
  TORCH_CHECK(attn_row.dim() == 2 && attn_row.size(0) == m && attn_row.size(1) == in_feat.size(1) &&
                  attn_col.sizes() == attn_row.sizes(),
              "attn_row / attn_col must have shape (", m, ", ", in_feat.size(1), "), got ", attn_row.sizes(), " / ",
              attn_col.sizes());
  TORCH_CHECK(in_feat.size(0) == m, "indptr describes ", m, " rows but in_feat has ", in_feat.size(0), " nodes");
  if (rows) {
    check_i32(*rows, "rows");
    check_edges(*rows, nnz, "rows");
    check_same_device(in_feat, {rows});
  }
  check_same_device(in_feat, {&attn_row, &attn_col, &indptr, &indices});
  return GatDims{(int)m, (int)nnz, (int)in_feat.size(1), (int)in_feat.size(2)};
}

// fused_gatconv.cpp:99-119
Tensor gat_hyper_fwd(const Tensor &attn_row, const Tensor &attn_col, const Tensor &indptr, const Tensor &indices,
                     const Tensor &rows, double slope, const Tensor &in_feat, int64_t plan, int64_t meta, bool need_ws) {
  const GatDims d = gat_checks(attn_row, attn_col, indptr, indices, &rows, in_feat);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(in_feat.device());
  Tensor out = torch::empty_like(in_feat), ws;
  if (need_ws) ws = torch::empty({d.h, d.nnz}, in_feat.options());
  check_rc(dfgnn_gat_hyper_fwd(d.m, d.nnz, d.h, d.f, indptr.data_ptr<int>(), indices.data_ptr<int>(), rows.data_ptr<int>(),
                               attn_row.data_ptr<float>(), attn_col.data_ptr<float>(), (float)slope,
                               in_feat.data_ptr<float>(), ws.defined() ? ws.data_ptr<float>() : nullptr,
                               out.data_ptr<float>(), plan_ptr(plan), plan_ptr(meta), cur_stream()),
           "gat_inference_hyper");
  return out;
}

// fused_gatconv.cpp:40-61 (use_lds) and :69-90 (global-memory logits)
Tensor gat_softmax_fwd(const Tensor &attn_row, const Tensor &attn_col, const Tensor &indptr, const Tensor &indices,
                       const Tensor &rows, double slope, const Tensor &in_feat, bool use_lds) {
  const GatDims d = gat_checks(attn_row, attn_col, indptr, indices, &rows, in_feat);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(in_feat.device());
  Tensor out = torch::empty_like(in_feat);
  Tensor logits = torch::empty({d.h, d.nnz}, in_feat.options());
  auto fn = use_lds ? dfgnn_gat_softmax_fwd : dfgnn_gat_softmax_gm_fwd;
  check_rc(fn(d.m, d.nnz, d.h, d.f, indptr.data_ptr<int>(), indices.data_ptr<int>(), rows.data_ptr<int>(),
              attn_row.data_ptr<float>(), attn_col.data_ptr<float>(), (float)slope, in_feat.data_ptr<float>(),
              logits.data_ptr<float>(), out.data_ptr<float>(), cur_stream()),
           use_lds ? "gat_inference_softmax" : "gat_inference_softmax_gm");
  return out;
}

// fused_gatconv.cpp:196-219
Tensor gat_tiling_fwd(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr, const Tensor &col_ind,
                      double slope, const Tensor &in_feat) {
  const GatDims d = gat_checks(attn_row, attn_col, row_ptr, col_ind, nullptr, in_feat);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(in_feat.device());
  Tensor out = torch::empty_like(in_feat);
  check_rc(dfgnn_gat_tiling_fwd(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(),
                                attn_row.data_ptr<float>(), attn_col.data_ptr<float>(), (float)slope,
                                in_feat.data_ptr<float>(), out.data_ptr<float>(), cur_stream()),
           "gat_inference_tiling");
  return out;
}

// fused_gatconv.cpp:11-32: the GAT training forward.  edge_mask: the dropout randoms [nnz, h] (undefined: no dropout);
// rows / plan / meta: the COO rows and the block plan when the batch may run on the matrix-core kernels, else undefined / 0
std::vector<Tensor> gat_fwd_train(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr, const Tensor &col_ind,
                                  const c10::optional<Tensor> &rows, double slope, const Tensor &in_feat,
                                  const c10::optional<Tensor> &edge_mask, double attn_drop, int64_t plan, int64_t meta) {
  const GatDims d = gat_checks(attn_row, attn_col, row_ptr, col_ind, rows ? &*rows : nullptr, in_feat);
  TORCH_CHECK(attn_drop >= 0.0 && attn_drop < 1.0, "attn_drop must be in [0, 1), got ", attn_drop);
  if (edge_mask) {
    check_f32(*edge_mask, "edge_mask");
    TORCH_CHECK(edge_mask->dim() == 2 && edge_mask->size(0) == d.nnz && edge_mask->size(1) == d.h, "edge_mask must have shape (",
                d.nnz, ", ", d.h, "), got ", edge_mask->sizes());
    check_same_device(in_feat, {&*edge_mask});
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(in_feat.device());
  Tensor out = torch::empty_like(in_feat);
  Tensor edge_max = torch::empty({d.m, d.h}, in_feat.options()), edge_sum = torch::empty({d.m, d.h}, in_feat.options());
  check_rc(dfgnn_gat_fwd_train(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(),
                               rows ? rows->data_ptr<int>() : nullptr, attn_row.data_ptr<float>(), attn_col.data_ptr<float>(),
                               (float)slope, in_feat.data_ptr<float>(), edge_mask ? edge_mask->data_ptr<float>() : nullptr,
                               (float)attn_drop, edge_max.data_ptr<float>(), edge_sum.data_ptr<float>(), out.data_ptr<float>(),
                               plan_ptr(plan), plan_ptr(meta), cur_stream()),
           "gat_forward");
  return {out, edge_max, edge_sum};
}

// fused_gatconv.cpp:291-353
std::vector<Tensor> gat_bwd(double slope, double attn_drop, const Tensor &row_ptr, const Tensor &col_ind,
                            const c10::optional<Tensor> &rows, const Tensor &col_ptr, const Tensor &row_ind, const Tensor &permute,
                            const Tensor &edge_max, const Tensor &edge_sum, const c10::optional<Tensor> &edge_mask,
                            const Tensor &in_feat, const Tensor &attn_row, const Tensor &attn_col, const Tensor &grad,
                            int64_t plan, int64_t meta) {
  const GatDims d = gat_checks(attn_row, attn_col, row_ptr, col_ind, rows ? &*rows : nullptr, in_feat);
  TORCH_CHECK(attn_drop >= 0.0 && attn_drop < 1.0, "attn_drop must be in [0, 1), got ", attn_drop);
  check_i32(col_ptr, "col_ptr");
  check_i32(row_ind, "row_ind");
  check_i32(permute, "permute");
  check_f32(edge_max, "edge_max");
  check_f32(edge_sum, "edge_sum");
  check_f32(grad, "grad");
  TORCH_CHECK(grad.sizes() == in_feat.sizes(), "grad has shape ", grad.sizes(), ", expected ", in_feat.sizes());
  TORCH_CHECK(edge_max.dim() == 2 && edge_max.size(0) == d.m && edge_max.size(1) == d.h && edge_sum.sizes() == edge_max.sizes(),
              "edge_max / edge_sum must have shape (", d.m, ", ", d.h, ")");
  TORCH_CHECK(col_ptr.size(0) == d.m + 1 && row_ind.size(0) == d.nnz && permute.size(0) == d.nnz,
              "col_ptr / row_ind / permute do not match the CSR structure");
  check_same_device(in_feat, {&col_ptr, &row_ind, &permute, &edge_max, &edge_sum, &grad});
  if (edge_mask) {
    check_f32(*edge_mask, "edge_mask");
    TORCH_CHECK(edge_mask->dim() == 2 && edge_mask->size(0) == d.nnz && edge_mask->size(1) == d.h, "edge_mask must have shape (",
                d.nnz, ", ", d.h, "), got ", edge_mask->sizes());
    check_same_device(in_feat, {&*edge_mask});
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(in_feat.device());
  Tensor grad_feat = torch::empty_like(in_feat);
  Tensor grad_row = torch::empty({d.m, d.h}, in_feat.options()), grad_col = torch::empty({d.m, d.h}, in_feat.options());
  Tensor grad_edge = torch::empty({d.h, d.nnz}, in_feat.options());
  check_rc(dfgnn_gat_bwd(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(),
                         rows ? rows->data_ptr<int>() : nullptr, col_ptr.data_ptr<int>(), row_ind.data_ptr<int>(),
                         permute.data_ptr<int>(), attn_row.data_ptr<float>(), attn_col.data_ptr<float>(), (float)slope,
                         in_feat.data_ptr<float>(), edge_max.data_ptr<float>(), edge_sum.data_ptr<float>(),
                         edge_mask ? edge_mask->data_ptr<float>() : nullptr, (float)attn_drop, grad.data_ptr<float>(),
                         grad_edge.data_ptr<float>(), grad_feat.data_ptr<float>(), grad_row.data_ptr<float>(),
                         grad_col.data_ptr<float>(), plan_ptr(plan), plan_ptr(meta), cur_stream()),
           "gat_backward");
  return {grad_feat, grad_row, grad_col};
}

// fused_gtconv.cpp:244-276 (tiling), :174-242 (csr, csr_gm), :316-389 (softmax, softmax_gm): the GT inference variants that
// take CSR (+ the COO rows for the two-kernel forms).  which: 0 tiling, 1 csr, 2 csr_gm, 3 softmax, 4 softmax_gm
Tensor gt_variant_fwd(int64_t which, const Tensor &indptr, const Tensor &indices, const c10::optional<Tensor> &rows, const Tensor &val,
                      const Tensor &Q, const Tensor &K, const Tensor &V, bool unit_val) {
  check_i32(indptr, "indptr");
  check_i32(indices, "indices");
  check_f32(val, "val");
  check_feat3(Q, Q, "Q");
  check_feat3(K, Q, "K");
  check_feat3(V, Q, "V");
  TORCH_CHECK(indptr.dim() == 1 && indices.dim() == 1, "indptr / indices must be 1-D");
  TORCH_CHECK(indptr.size(0) - 1 == Q.size(0), "indptr describes ", indptr.size(0) - 1, " rows but features have ", Q.size(0), " nodes");
  const int m = (int)Q.size(0), nnz = (int)indices.size(0), h = (int)Q.size(1), f = (int)Q.size(2);
  check_edges(val, nnz, "val");
  check_same_device(Q, {&indptr, &indices, &val, &K, &V});
  const bool needs_rows = which >= 3;
  TORCH_CHECK(which >= 0 && which <= 4, "unknown GT variant ", which);
  if (needs_rows) {
    TORCH_CHECK(rows.has_value(), "rows is required by the softmax variants");
    check_i32(*rows, "rows");
    check_edges(*rows, nnz, "rows");
    check_same_device(Q, {&*rows});
  }
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor out = torch::empty_like(Q), logits;
  if (which != 0) logits = torch::empty({h, nnz}, Q.options());
  const int *ip = indptr.data_ptr<int>(), *ci = indices.data_ptr<int>();
  const float *q = Q.data_ptr<float>(), *k = K.data_ptr<float>(), *v = V.data_ptr<float>();
  float *lg = logits.defined() ? logits.data_ptr<float>() : nullptr, *o = out.data_ptr<float>();
  // (the two-kernel 'softmax' forms multiply the values in as they are; the others take NULL for all ones)
  const float *vl = (which >= 3 || !unit_val) ? val.data_ptr<float>() : nullptr;
  int rc = 0;
  switch (which) {
    case 0: rc = dfgnn_gt_tiling_fwd(m, nnz, h, f, ip, ci, val.data_ptr<float>(), q, k, v, o, cur_stream()); break;
    case 1: rc = dfgnn_gt_csr_fwd(m, nnz, h, f, ip, ci, vl, q, k, v, lg, o, cur_stream()); break;
    case 2: rc = dfgnn_gt_csr_gm_fwd(m, nnz, h, f, ip, ci, vl, q, k, v, lg, o, cur_stream()); break;
    case 3: rc = dfgnn_gt_softmax_fwd(m, nnz, h, f, ip, ci, rows->data_ptr<int>(), vl, q, k, v, lg, o, cur_stream()); break;
    default: rc = dfgnn_gt_softmax_gm_fwd(m, nnz, h, f, ip, ci, rows->data_ptr<int>(), vl, q, k, v, lg, o, cur_stream()); break;
  }
  static const char *names[] = {"gt_tiling_inference", "gt_csr_inference", "gt_csr_gm_inference", "gt_softmax_inference",
                                "gt_softmax_gm_inference"};
  check_rc(rc, names[which]);
  return out;
}

// dfgnn_plan_build: -> (plan buffer int32[dfgnn_plan_ints], its 12 header words); synchronises the current stream once
std::pair<Tensor, std::vector<int64_t>> plan_build(const Tensor &indptr, const Tensor &indices, int64_t f) {
  check_i32(indptr, "indptr");
  check_i32(indices, "indices");
  TORCH_CHECK(indptr.dim() == 1 && indices.dim() == 1 && indptr.size(0) >= 1, "indptr / indices must be 1-D");
  check_same_device(indptr, {&indices});
  const int m = (int)indptr.size(0) - 1, nnz = (int)indices.size(0);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(indptr.device());
  Tensor buf = torch::empty({(int64_t)dfgnn_plan_ints(m, nnz)}, indptr.options());
  int meta[12];
  check_rc(dfgnn_plan_build(m, nnz, (int)f, indptr.data_ptr<int>(), indices.data_ptr<int>(), buf.data_ptr<int>(), meta,
                            cur_stream()),
           "dfgnn_plan_build");
  return {buf, std::vector<int64_t>(meta, meta + 12)};
}

// dfgnn_preprocess_hyper: COO -> (row_ptr, col_ind, rows, edge_order[, col_ptr, row_ind, val_idx]) (DFGNN/layers/util.py:82-142)
std::vector<Tensor> preprocess_hyper(const Tensor &src, const Tensor &dst, int64_t num_nodes, bool csc) {
  TORCH_CHECK(src.is_cuda() && dst.is_cuda(), "src / dst must be on CUDA");
  TORCH_CHECK(src.scalar_type() == dst.scalar_type() && (src.scalar_type() == torch::kInt64 || src.scalar_type() == torch::kInt32),
              "src / dst must both be int64 or int32, got ", src.scalar_type(), " / ", dst.scalar_type());
  TORCH_CHECK(src.dim() == 1 && src.sizes() == dst.sizes(), "src / dst must be 1-D and of equal length, got ", src.sizes(), " / ",
              dst.sizes());
  check_same_device(src, {&dst});
  const Tensor s = src.contiguous(), t = dst.contiguous();
  const int64_t nnz = s.numel();
  TORCH_CHECK(nnz < (int64_t(1) << 31) && num_nodes < (int64_t(1) << 31) && num_nodes >= 0,
              "graphs with 2^31 or more nodes / edges are not supported (int32 index arrays)");
  const int m = (int)num_nodes;
  c10::hip::HIPGuardMasqueradingAsCUDA guard(s.device());
  const auto i32 = s.options().dtype(torch::kInt32);
  std::vector<Tensor> outs = {torch::empty({m + 1}, i32), torch::empty({nnz}, i32), torch::empty({nnz}, i32), torch::empty({nnz}, i32)};
  if (csc) {
    outs.push_back(torch::empty({m + 1}, i32));
    outs.push_back(torch::empty({nnz}, i32));
    outs.push_back(torch::empty({nnz}, i32));
  }
  const size_t ws_bytes = dfgnn_preprocess_ws_bytes(m, (int)nnz);
  Tensor ws = torch::empty({(int64_t)ws_bytes}, s.options().dtype(torch::kUInt8));
  check_rc(dfgnn_preprocess_hyper(m, (int)nnz, s.data_ptr(), t.data_ptr(), s.scalar_type() == torch::kInt64 ? 1 : 0,
                                  outs[0].data_ptr<int>(), outs[1].data_ptr<int>(), outs[2].data_ptr<int>(), outs[3].data_ptr<int>(),
                                  csc ? outs[4].data_ptr<int>() : nullptr, csc ? outs[5].data_ptr<int>() : nullptr,
                                  csc ? outs[6].data_ptr<int>() : nullptr, ws.data_ptr(), ws_bytes, cur_stream()),
           "dfgnn_preprocess_hyper");
  return outs;
}

}  // namespace

PYBIND11_MODULE(_dfgnn_ext, m) {
  m.doc() = "torch C++ binding of libdfgnn.so (include/dfgnn.h); see df-gnn_amd/fused_gtconv.py / fused_gatconv.py";
  // compile-time constants of THIS extension (not the library's answers: comparing those with the library would compare
  // the library with itself): dfgnn_native.ext() takes the extension only if both equal the library's
  m.def("abi_version", [] { return (int)DFGNN_ABI_VERSION; });
  m.def("build_id", [] { return std::string(DFGNN_SRC_HASH); });
  m.def("gt_hyper_fwd", &gt_hyper_fwd, "fused GT conv 'hyper' forward (inference / training)");
  m.def("gt_bwd", &gt_bwd, "fused GT conv backward");
  m.def("gt_hyper_fwd_stats", &gt_hyper_fwd_stats, "fused GT conv 'hyper' training forward, row statistics instead of attn_edge");
  m.def("gt_bwd_stats", &gt_bwd_stats, "fused GT conv backward from the row statistics");
  m.def("gt_hyper_fwd_ranked", &gt_hyper_fwd_ranked, "fused GT conv 'hyper' training forward, attention values in rank order");
  m.def("gt_bwd_ranked", &gt_bwd_ranked, "fused GT conv backward from rank-ordered attention values");
  m.def("plan_dense_weights", &plan_dense_weights, "edge values of a plan's dense ranges in dense form (dfgnn_plan_dense_weights)");
  m.def("gat_hyper_fwd", &gat_hyper_fwd, "fused GAT conv 'hyper' inference");
  m.def("gat_softmax_fwd", &gat_softmax_fwd, "fused GAT conv 'softmax' / 'softmax_gm' inference");
  m.def("gat_tiling_fwd", &gat_tiling_fwd, "fused GAT conv 'tiling' inference");
  m.def("gat_fwd_train", &gat_fwd_train, "fused GAT conv training forward (row statistics, attention dropout)");
  m.def("gat_bwd", &gat_bwd, "fused GAT conv backward");
  m.def("gt_variant_fwd", &gt_variant_fwd, "fused GT conv inference: tiling / csr / csr_gm / softmax / softmax_gm");
  m.def("plan_build", &plan_build, "block plan of a CSR structure (dfgnn_plan_build)");
  m.def("preprocess_hyper", &preprocess_hyper, "COO -> CSR / COO rows / CSC on the GPU (dfgnn_preprocess_hyper)");
}
