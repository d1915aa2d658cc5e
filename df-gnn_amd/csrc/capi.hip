// capi.hip -- the extern "C" boundary of libdfgnn.so (declared in include/dfgnn.h).
// Argument validation + dispatch only; no allocation, no synchronisation.
#include "../../include/dfgnn.h"
#include "dfgnn_launch.hpp"
#include "dfgnn_errstr.h"

#include <mutex>
#include <set>
#include <utility>

using namespace dfgnn;

namespace dfgnn {
int set_max_lds_cached_ptr(const void *fn) {
  static std::mutex mu;
  static std::set<std::pair<int, const void *>> done;
  int dev = 0;
  if (hipError_t rc = hipGetDevice(&dev)) return (int)rc;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, fn})) return 0;
  if (hipError_t rc = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes)) return (int)rc;
  done.insert({dev, fn});
  return 0;
}
}  // namespace dfgnn

namespace {
inline hipStream_t as_stream(dfgnn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// A plan is used only if it was built for exactly this (m, nnz, f) and LDS budget; otherwise the call
// silently takes the general kernels (same results, no LDS residency).
// The plan kernels (LDS-resident and matrix-core) address a feature row with 32-bit byte offsets; the general kernels
// use size_t throughout.  Feature matrices of 4 GiB or more therefore take the general kernels.
inline bool plan_offsets_fit(int m, int h, int f) { return (size_t)m * (size_t)h * (size_t)f * 4u < (1ull << 32); }

inline bool make_plan(Plan &p, const int *plan_dev, const int *meta, int m, int nnz, int h, int f) {
  p = Plan{nullptr, 0, 0, 0, 0, m, nnz, f, 0, 0};
  if (!plan_dev || !meta) return false;
  if (!plan_offsets_fit(m, h, f)) return false;
  if (meta[4] != m || meta[5] != nnz || meta[6] != f || meta[7] != kBlockLdsBudget) return false;
  if (meta[0] <= 0) return false;
  // Low-degree batches (e.g. molecule / peptide graphs, ~2 edges per row) are bound by the node features, not
  // by the per-edge gathers: a 1024-thread workgroup per range only adds fixed cost there (measured on the
  // Peptides-like config: 279 us resident vs 180 us general for fwd+bwd), so such graphs keep the general kernels.
  if ((long)nnz < (long)kBlockMinAvgDegree * m) return false;
  p = Plan{plan_dev, meta[0], meta[1], meta[2], meta[3], m, nnz, f, meta[8], meta[9], meta[10], meta[11]};
  if (p.num_dense > 0 && p.coords_off <= 0) return false;  // (a plan of an older layout)
  return true;
}

// A built plan is usable for width f when the LDS-resident kernels have an exact lane layout for f, or -- f = 8, which
// has none -- when every fit range goes to the matrix-core kernels anyway (unit edge values, COO rows, not disabled).
inline bool plan_usable(const Plan &p, int f, bool unit_val, const int *rows) {
  if (block_width_ok(f)) return true;
  return f == 8 && unit_val && rows && dense_enabled() && p.num_dense == p.num_fit;
}

// Returns <0 on a bad argument, 1 when there is nothing to do, 0 to proceed.
inline int check_common(int m, int nnz, int h, int f, const void *row_ptr, const void *col_ind) {
  if (m < 0 || nnz < 0 || h < 0 || f < 0) return kErrBadArg;
  if (m == 0 || h == 0 || f == 0) return 1;
  if (!row_ptr) return kErrBadArg;
  if (nnz > 0 && !col_ind) return kErrBadArg;
  if (h > 65535) return kErrUnsupported;
  return 0;
}
}  // namespace

extern "C" {

int dfgnn_abi_version(void) { return DFGNN_ABI_VERSION; }

int dfgnn_plan_applies(int m, int nnz, int h, int f, const int *plan_meta) {
  Plan p;
  static const int token = 0;  // (make_plan only stores the device pointer)
  return make_plan(p, &token, plan_meta, m, nnz, h, f) ? 1 : 0;
}

#ifndef DFGNN_SRC_HASH
#define DFGNN_SRC_HASH "unknown"
#endif
const char *dfgnn_build_id(void) { return DFGNN_SRC_HASH; }

const char *dfgnn_error_string(int code) {
  if (const char *text = dfgnn_static_error_string(code)) return text;
  return hipGetErrorString(static_cast<hipError_t>(code));
}

int dfgnn_gt_hyper_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                       const float *val, const float *Q, const float *K, const float *V, float *attn_edge,
                       float *edge_ws, float *out, const int *plan, const int *plan_meta,
                       dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !out || (nnz > 0 && !rows)) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, val};
  Plan p;
  const bool v4 = (f % 4 == 0) && aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(out);
  if (v4 && make_plan(p, plan, plan_meta, m, nnz, h, f) && plan_usable(p, f, !val, rows)) {
    if (int rc = launch_gt_block_fwd(g, p, Q, K, V, attn_edge, edge_ws, out, as_stream(stream))) return rc;
    return launch_gt_hyper_fwd(g, Q, K, V, attn_edge, out, p.spill(), p.num_spill, as_stream(stream));
  }
  if (low_degree(m, nnz)) return launch_gt_lowdeg_fwd(g, Q, K, V, attn_edge, out, as_stream(stream));
  return launch_gt_hyper_fwd(g, Q, K, V, attn_edge, out, nullptr, 0, as_stream(stream));
}

int dfgnn_gt_bwd_rows(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                      const float *val, const float *K, const float *V, const float *attn_edge,
                      const float *grad_out, float *grad_edge, float *dQ, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!K || !V || !grad_out || !dQ) return kErrBadArg;
  if (nnz > 0 && (!rows || !attn_edge || !grad_edge)) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, val};
  return launch_gt_bwd_rows(g, K, V, attn_edge, grad_out, grad_edge, dQ, nullptr, 0, as_stream(stream));
}

int dfgnn_gt_bwd_cols(int m, int nnz, int h, int f, const float *val, const int *col_ptr, const int *row_ind,
                      const int *val_idx, const float *Q, const float *attn_edge, const float *grad_edge,
                      const float *grad_out, float *dK, float *dV, dfgnn_stream_t stream) {
  if (m < 0 || nnz < 0 || h < 0 || f < 0) return kErrBadArg;
  if (m == 0 || h == 0 || f == 0) return 0;
  if (h > 65535) return kErrUnsupported;
  if (!col_ptr || !Q || !grad_out || !dK || !dV) return kErrBadArg;
  if (nnz > 0 && (!row_ind || !val_idx || !attn_edge || !grad_edge)) return kErrBadArg;
  const Csr g{m, nnz, h, f, nullptr, nullptr, nullptr, val};
  return launch_gt_bwd_cols(g, col_ptr, row_ind, val_idx, Q, attn_edge, grad_edge, grad_out, dK, dV, nullptr, 0,
                            as_stream(stream));
}

int dfgnn_gt_bwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                 const float *val, const int *col_ptr, const int *row_ind, const int *val_idx, const float *Q,
                 const float *K, const float *V, const float *attn_edge, const float *grad_out,
                 float *grad_edge, float *dQ, float *dK, float *dV, const int *plan, const int *plan_meta,
                 dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !grad_out || !dQ || !dK || !dV || !col_ptr) return kErrBadArg;
  if (nnz > 0 && (!rows || !row_ind || !val_idx || !attn_edge || !grad_edge)) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, val};
  Plan p;
  const bool v4 = (f % 4 == 0) && aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(grad_out) &&
                  aligned16(dQ) && aligned16(dK) && aligned16(dV);
  const int *chunks = nullptr;
  int nchunks = 0;
  if (v4 && make_plan(p, plan, plan_meta, m, nnz, h, f) && plan_usable(p, f, !val, rows)) {
    if (int rc = launch_gt_block_bwd(g, p, col_ptr, row_ind, val_idx, Q, K, V, attn_edge, grad_out, grad_edge, dQ,
                                     dK, dV, as_stream(stream)))
      return rc;
    if (p.num_spill == 0) return 0;
    chunks = p.spill();
    nchunks = p.num_spill;
  }
  if (!chunks && low_degree(m, nnz))
    return launch_gt_lowdeg_bwd(g, col_ptr, row_ind, val_idx, Q, K, V, attn_edge, grad_out, grad_edge, dQ, dK, dV,
                                as_stream(stream));
  if (int rc = launch_gt_bwd_rows(g, K, V, attn_edge, grad_out, grad_edge, dQ, chunks, nchunks, as_stream(stream)))
    return rc;
  return launch_gt_bwd_cols(g, col_ptr, row_ind, val_idx, Q, attn_edge, grad_edge, grad_out, dK, dV, chunks,
                            nchunks, as_stream(stream));
}

// ---- the statistics-saving training pair (gt_dense_stats.hip) -----------------------------------------------------------
// usable when every range of the plan is served by the matrix-core kernels (unit edge values are the caller's promise)
static bool gt_stats_plan(Plan &p, const int *plan, const int *plan_meta, int m, int nnz, int h, int f) {
  if (!dense_enabled()) return false;
  if (f != 8 && f != 16 && f != 32 && f != 64 && f != 128) return false;
  if (!make_plan(p, plan, plan_meta, m, nnz, h, f)) return false;
  return p.num_dense > 0 && p.num_dense == p.num_fit && p.num_spill == 0;
}

int dfgnn_gt_stats_applies(int m, int nnz, int h, int f, const int *plan_meta) {
  Plan p;
  static const int token = 0;
  return gt_stats_plan(p, &token, plan_meta, m, nnz, h, f) ? 1 : 0;
}

int dfgnn_gt_hyper_fwd_stats(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *weights,
                             const float *Q, const float *K, const float *V, float *row_max, float *row_sum, float *out,
                             const int *plan, const int *plan_meta, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !out || (!row_max != !row_sum)) return kErrBadArg;  // (both statistics or neither: inference)
  Plan p;
  if (!(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(out)) || !gt_stats_plan(p, plan, plan_meta, m, nnz, h, f))
    return kErrUnsupported;
  if (weights && !aligned16(weights)) return kErrUnsupported;
  Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, nullptr};
  g.wdense = weights;
  return launch_gt_dense_fwd_stats(g, p, Q, K, V, out, row_max, row_sum, as_stream(stream));
}

int dfgnn_gt_bwd_stats(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *weights,
                       const float *Q, const float *K, const float *V, const float *row_max, const float *row_sum,
                       const float *grad_out, float *dQ, float *dK, float *dV, const int *plan, const int *plan_meta,
                       dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !row_max || !row_sum || !grad_out || !dQ || !dK || !dV) return kErrBadArg;
  Plan p;
  if (!(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(grad_out) && aligned16(dQ) && aligned16(dK) && aligned16(dV)) ||
      !gt_stats_plan(p, plan, plan_meta, m, nnz, h, f))
    return kErrUnsupported;
  if (weights && !aligned16(weights)) return kErrUnsupported;
  Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, nullptr};
  g.wdense = weights;
  return launch_gt_dense_bwd_stats(g, p, Q, K, V, row_max, row_sum, grad_out, dQ, dK, dV, as_stream(stream));
}

// the attn_edge pair in rank order (all-dense plan, unit edge values)
int dfgnn_gt_hyper_fwd_ranked(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *Q,
                              const float *K, const float *V, float *attn_ranked, float *out, const int *plan,
                              const int *plan_meta, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !out || !attn_ranked) return kErrBadArg;
  Plan p;
  if (!(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(out)) || !gt_stats_plan(p, plan, plan_meta, m, nnz, h, f))
    return kErrUnsupported;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, nullptr};
  return launch_gt_dense_fwd_ranked(g, p, Q, K, V, attn_ranked, out, as_stream(stream));
}

int dfgnn_gt_bwd_ranked(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *Q,
                        const float *K, const float *V, const float *attn_ranked, const float *grad_out, float *dQ,
                        float *dK, float *dV, const int *plan, const int *plan_meta, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !attn_ranked || !grad_out || !dQ || !dK || !dV) return kErrBadArg;
  Plan p;
  if (!(aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(grad_out) && aligned16(dQ) && aligned16(dK) && aligned16(dV)) ||
      !gt_stats_plan(p, plan, plan_meta, m, nnz, h, f))
    return kErrUnsupported;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, nullptr};
  return launch_gt_dense_bwd(g, p, Q, K, V, attn_ranked, grad_out, dQ, dK, dV, as_stream(stream), true);
}

int dfgnn_gt_tiling_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *val,
                        const float *Q, const float *K, const float *V, float *out, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !out) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, val};
  return launch_gt_tiling_fwd(g, Q, K, V, out, as_stream(stream));
}

static int gt_csr_impl(bool use_lds, int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *val,
                       const float *Q, const float *K, const float *V, float *logits, float *out, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !out || (nnz > 0 && !logits)) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, val};
  return launch_gt_csr_fwd(g, Q, K, V, logits, out, use_lds, as_stream(stream));
}

int dfgnn_gt_csr_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *val,
                     const float *Q, const float *K, const float *V, float *logits, float *out, dfgnn_stream_t stream) {
  return gt_csr_impl(true, m, nnz, h, f, row_ptr, col_ind, val, Q, K, V, logits, out, stream);
}

int dfgnn_gt_csr_gm_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *val,
                        const float *Q, const float *K, const float *V, float *logits, float *out,
                        dfgnn_stream_t stream) {
  return gt_csr_impl(false, m, nnz, h, f, row_ptr, col_ind, val, Q, K, V, logits, out, stream);
}

int dfgnn_gat_recompute_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *attn_row,
                            const float *attn_col, float negative_slope, const float *X, float *out,
                            dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!attn_row || !attn_col || !X || !out) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, nullptr};
  return launch_gat_recompute_fwd(g, attn_row, attn_col, negative_slope, X, out, as_stream(stream));
}

static int gt_softmax_impl(bool use_lds, int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                           const int *rows, const float *val, const float *Q, const float *K, const float *V,
                           float *logits, float *out, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!Q || !K || !V || !out || (nnz > 0 && (!rows || !logits))) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, val};
  if (int rc = launch_gt_sddmm(g, Q, K, logits, as_stream(stream))) return rc;
  return launch_softmax_spmm(g, logits, V, out, use_lds, as_stream(stream));
}

int dfgnn_gt_softmax_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                         const float *val, const float *Q, const float *K, const float *V, float *logits,
                         float *out, dfgnn_stream_t stream) {
  return gt_softmax_impl(true, m, nnz, h, f, row_ptr, col_ind, rows, val, Q, K, V, logits, out, stream);
}

int dfgnn_gt_softmax_gm_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                            const int *rows, const float *val, const float *Q, const float *K, const float *V,
                            float *logits, float *out, dfgnn_stream_t stream) {
  return gt_softmax_impl(false, m, nnz, h, f, row_ptr, col_ind, rows, val, Q, K, V, logits, out, stream);
}

int dfgnn_gat_hyper_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                        const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                        float *edge_ws, float *out, const int *plan, const int *plan_meta, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!attn_row || !attn_col || !X || !out || (nnz > 0 && !rows)) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, nullptr};
  Plan p;
  const bool v4 = (f % 4 == 0) && aligned16(X) && aligned16(out);
  if (v4 && make_plan(p, plan, plan_meta, m, nnz, h, f) && plan_usable(p, f, true, rows)) {
    if (int rc = launch_gat_block_fwd(g, p, attn_row, attn_col, negative_slope, X, edge_ws, out, as_stream(stream)))
      return rc;
    return launch_gat_hyper_fwd(g, attn_row, attn_col, negative_slope, X, out, p.spill(), p.num_spill,
                                as_stream(stream));
  }
  return launch_gat_hyper_fwd(g, attn_row, attn_col, negative_slope, X, out, nullptr, 0, as_stream(stream));
}

static int gat_softmax_impl(bool use_lds, int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                            const int *rows, const float *attn_row, const float *attn_col, float negative_slope,
                            const float *X, float *logits, float *out, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!attn_row || !attn_col || !X || !out || (nnz > 0 && (!rows || !logits))) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, nullptr};
  if (int rc = launch_gat_sddmm(g, attn_row, attn_col, negative_slope, logits, as_stream(stream))) return rc;
  return launch_softmax_spmm(g, logits, X, out, use_lds, as_stream(stream));
}

int dfgnn_gat_softmax_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                          const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                          float *logits, float *out, dfgnn_stream_t stream) {
  return gat_softmax_impl(true, m, nnz, h, f, row_ptr, col_ind, rows, attn_row, attn_col, negative_slope, X,
                          logits, out, stream);
}

int dfgnn_gat_softmax_gm_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                             const int *rows, const float *attn_row, const float *attn_col, float negative_slope,
                             const float *X, float *logits, float *out, dfgnn_stream_t stream) {
  return gat_softmax_impl(false, m, nnz, h, f, row_ptr, col_ind, rows, attn_row, attn_col, negative_slope, X,
                          logits, out, stream);
}

int dfgnn_gat_tiling_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                         const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                         float *out, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!attn_row || !attn_col || !X || !out) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, nullptr, nullptr};
  return launch_gat_tiling_fwd(g, attn_row, attn_col, negative_slope, X, out, as_stream(stream));
}

int dfgnn_gat_attn_scores(int m, int h, int f, const float *a_l, const float *a_r, const float *X, float *attn_row,
                          float *attn_col, dfgnn_stream_t stream) {
  if (m < 0 || h < 0 || f < 0) return kErrBadArg;
  if (m == 0 || h == 0) return 0;
  if (h > 65535) return kErrUnsupported;
  if (!a_l || !a_r || !X || !attn_row || !attn_col || f == 0) return kErrBadArg;
  return launch_gat_attn_scores(m, h, f, a_l, a_r, X, attn_row, attn_col, as_stream(stream));
}

// The GAT training pair serves the dense ranges of a plan with the matrix-core kernels and everything else (its
// non-dense fit ranges, its spill chunks) with the general kernels restricted to those ranges.
static bool gat_train_dense(Plan &p, const int *rows, const int *plan, const int *plan_meta, int m, int nnz, int h,
                            int f, bool v4) {
  if (!rows || !v4 || !dense_enabled()) return false;
  if (f != 8 && f != 16 && f != 32 && f != 64 && f != 128) return false;
  if (!make_plan(p, plan, plan_meta, m, nnz, h, f)) return false;
  return p.num_dense > 0;
}
static bool plan_has_rest(const Plan &p) { return p.num_fit - p.num_dense + p.num_spill > 0; }

int dfgnn_gat_fwd_train(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                        const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                        const float *edge_mask, float attn_drop, float *edge_max, float *edge_sum, float *out,
                        const int *plan, const int *plan_meta, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!attn_row || !attn_col || !X || !out || !edge_max || !edge_sum) return kErrBadArg;
  if (edge_mask && !(attn_drop >= 0.f && attn_drop < 1.f)) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, nullptr};
  Plan p;
  const bool v4 = (f % 4 == 0) && aligned16(X) && aligned16(out);
  if (gat_train_dense(p, rows, plan, plan_meta, m, nnz, h, f, v4)) {
    if (int rc = launch_gat_dense_fwd(g, p, attn_row, attn_col, negative_slope, X, out, as_stream(stream), edge_max,
                                      edge_sum, edge_mask, attn_drop))
      return rc;
    if (!plan_has_rest(p)) return 0;
    return launch_gat_train_fwd(g, attn_row, attn_col, negative_slope, X, edge_mask, attn_drop, edge_max, edge_sum,
                                out, as_stream(stream), &p);
  }
  return launch_gat_train_fwd(g, attn_row, attn_col, negative_slope, X, edge_mask, attn_drop, edge_max, edge_sum,
                              out, as_stream(stream));
}

int dfgnn_gat_bwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const int *rows,
                  const int *col_ptr, const int *row_ind, const int *permute, const float *attn_row,
                  const float *attn_col, float negative_slope, const float *X, const float *edge_max,
                  const float *edge_sum, const float *edge_mask, float attn_drop, const float *grad_out,
                  float *grad_edge, float *grad_feat, float *grad_attn_row, float *grad_attn_col, const int *plan,
                  const int *plan_meta, dfgnn_stream_t stream) {
  if (int c = check_common(m, nnz, h, f, row_ptr, col_ind)) return c < 0 ? c : 0;
  if (!attn_row || !attn_col || !X || !edge_max || !edge_sum || !grad_out || !grad_feat || !grad_attn_row ||
      !grad_attn_col)
    return kErrBadArg;
  if (edge_mask && !(attn_drop >= 0.f && attn_drop < 1.f)) return kErrBadArg;
  const Csr g{m, nnz, h, f, row_ptr, col_ind, rows, nullptr};
  Plan p;
  const bool v4 = (f % 4 == 0) && aligned16(X) && aligned16(grad_out) && aligned16(grad_feat);
  const Plan *rest = nullptr;
  if (gat_train_dense(p, rows, plan, plan_meta, m, nnz, h, f, v4)) {
    if (int rc = launch_gat_dense_bwd(g, p, attn_row, attn_col, negative_slope, X, edge_max, edge_sum, grad_out,
                                      grad_feat, grad_attn_row, grad_attn_col, as_stream(stream), edge_mask, attn_drop))
      return rc;
    if (!plan_has_rest(p)) return 0;
    rest = &p;
  }
  if (!col_ptr || (nnz > 0 && (!row_ind || !permute || !grad_edge))) return kErrBadArg;
  if (int rc = launch_gat_bwd_rows(g, attn_row, attn_col, negative_slope, X, edge_max, edge_sum, edge_mask,
                                   attn_drop, grad_out, grad_edge, grad_attn_row, as_stream(stream), rest))
    return rc;
  return launch_gat_bwd_cols(g, col_ptr, row_ind, permute, attn_row, attn_col, negative_slope, edge_max, edge_sum,
                             edge_mask, attn_drop, grad_edge, grad_out, grad_feat, grad_attn_col, as_stream(stream),
                             rest);
}

}  // extern "C"
