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

std::vector<Tensor> gt_hyper_fwd_stats(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &Q, const Tensor &K,
                                       const Tensor &V, int64_t plan, int64_t meta) {
  const GtDims d = gt_stats_checks(row_ptr, col_ind, Q, K, V);
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Q.device());
  Tensor out = torch::empty_like(Q);
  Tensor row_max = torch::empty({d.m, d.h}, Q.options()), row_sum = torch::empty({d.m, d.h}, Q.options());
  check_rc(dfgnn_gt_hyper_fwd_stats(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), Q.data_ptr<float>(),
                                    K.data_ptr<float>(), V.data_ptr<float>(), row_max.data_ptr<float>(),
                                    row_sum.data_ptr<float>(), out.data_ptr<float>(), plan_ptr(plan), plan_ptr(meta),
                                    cur_stream()),
           "gt_hyper_forward_stats");
  return {out, row_max, row_sum};
}

std::vector<Tensor> gt_bwd_stats(const Tensor &row_ptr, const Tensor &col_ind, const Tensor &Q, const Tensor &K,
                                 const Tensor &V, const Tensor &row_max, const Tensor &row_sum, const Tensor &grad,
                                 int64_t plan, int64_t meta) {
  const GtDims d = gt_stats_checks(row_ptr, col_ind, Q, K, V);
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
  check_rc(dfgnn_gt_bwd_stats(d.m, d.nnz, d.h, d.f, row_ptr.data_ptr<int>(), col_ind.data_ptr<int>(), Q.data_ptr<float>(),
                              K.data_ptr<float>(), V.data_ptr<float>(), row_max.data_ptr<float>(), row_sum.data_ptr<float>(),
                              grad.data_ptr<float>(), dQ.data_ptr<float>(), dK.data_ptr<float>(), dV.data_ptr<float>(),
                              plan_ptr(plan), plan_ptr(meta), cur_stream()),
           "gt_backward_stats");
  return {dQ, dK, dV};
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
  }
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

}  // namespace

PYBIND11_MODULE(_dfgnn_ext, m) {
  m.doc() = "torch C++ binding of libdfgnn.so (include/dfgnn.h); see df-gnn_amd/fused_gtconv.py / fused_gatconv.py";
  m.def("abi_version", [] { return dfgnn_abi_version(); });
  m.def("build_id", [] { return std::string(dfgnn_build_id()); });
  m.def("gt_hyper_fwd", &gt_hyper_fwd, "fused GT conv 'hyper' forward (inference / training)");
  m.def("gt_bwd", &gt_bwd, "fused GT conv backward");
  m.def("gt_hyper_fwd_stats", &gt_hyper_fwd_stats, "fused GT conv 'hyper' training forward, row statistics instead of attn_edge");
  m.def("gt_bwd_stats", &gt_bwd_stats, "fused GT conv backward from the row statistics");
  m.def("gat_hyper_fwd", &gat_hyper_fwd, "fused GAT conv 'hyper' inference");
  m.def("gat_softmax_fwd", &gat_softmax_fwd, "fused GAT conv 'softmax' / 'softmax_gm' inference");
  m.def("gat_tiling_fwd", &gat_tiling_fwd, "fused GAT conv 'tiling' inference");
}
