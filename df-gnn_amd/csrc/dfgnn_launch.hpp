// dfgnn_launch.hpp -- host-side plumbing shared by the launcher translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "dfgnn_device.hpp"

namespace dfgnn {

struct Csr {
  int m, nnz, h, f;
  const int *row_ptr;
  const int *col_ind;
  const int *rows;   // sorted COO row ids (hyper / softmax formats), may be null for CSR-only ops
  const float *val;  // may be null (all ones)
  // matrix-core kernels only (set by their launchers from the plan): uint16 per edge, (row << 8 | column) within the
  // edge's dense range -- read instead of rows / col_ind
  const unsigned short *coords = nullptr;
  // edge bitmaps of the dense ranges (plan.hip: masks), 8 words per node: bit c of mask[8 i ..] = node i has an edge to
  // the node c places after the first node of its range; maskT likewise for the edges INTO node i.  Read by the
  // statistics-saving training pair (gt_dense_stats.hip), which needs to know where the edges are but not their order.
  const unsigned *mask = nullptr, *maskT = nullptr;
  // edge VALUES of the dense ranges in dense form (plan.hip: plan_dense_weights), kPlanWeightStride floats per node:
  // wdense[256 i + c] = val of the edge from node i to the node c places after the first node of its range.  Read by the
  // WEIGHTED instances of the statistics-saving pair (gt_dense_stats_w.hip) wherever the edge bitmap has a bit; null = unit values
  const float *wdense = nullptr;
};
constexpr int kPlanWeightStride = 256;

constexpr int kErrBadArg = -1;
constexpr int kErrUnsupported = -2;

// Rows handled by one workgroup of the hyper-format kernels, and the LDS budget (floats) for the
// workgroup's edge logits.  A workgroup whose rows hold more edges than kHyperCap does not use the
// LDS path: its waves run the online (tiled) row routine instead, so there is no degree limit
// (the reference hard-codes 128 neighbours/row and overflows).
constexpr int kHyperRows = 16;
constexpr int kHyperCap = 4096;
// per-wave scratch for the online routine: 64 weights + 64 column ids
constexpr int kScratchFloatsPerWave = 2 * kWave;

// ---- block plan (plan.hip) ---------------------------------------------------------------------
// One workgroup of kBlockThreads keeps a closed node range's feature rows resident in LDS.
constexpr int kBlockThreads = 1024;                      // 16 waves, 4 per SIMD
constexpr int kBlockWaves = kBlockThreads / kWave;
constexpr int kBlockScratchBytes = kBlockWaves * kWave * 8;  // per-wave (col, weight) staging, 512 B each
constexpr int kLdsBytes = 160 * 1024;
constexpr int kBlockLdsBudget = kLdsBytes - kBlockScratchBytes - 256;  // resident rows + edge values + 1/sum
constexpr int kBlockMinAvgDegree = 8;                    // below this average degree the general kernels win
constexpr int kBlockMergeNodes = 256;                    // small graphs are merged up to this many nodes

constexpr int kPlanHeader = 12;                          // int32 header words of a plan buffer
constexpr int kPlanEdgeGlobal = 1 << 30;                 // flag on a fit range's n1: per-edge array in global scratch
constexpr int kPlanDense = 1 << 29;                      // flag on a fit range's n1: qualifies for the matrix-core kernels
constexpr int kPlanRangeMask = ~(kPlanEdgeGlobal | kPlanDense);

constexpr int kPlanMaskWords = 8;                        // 32-bit words of a node's edge bitmap (dense ranges: < 256 nodes)
// int32 offset of the edge bitmaps in a plan buffer whose packed coordinates start at coords_off
inline size_t plan_mask_off(size_t coords_off, int nnz) { return (coords_off + ((size_t)nnz + 1) / 2 + 4 + 3) & ~(size_t)3; }

struct Plan {              // host view of a built plan (see plan.hip for the device layout)
  const int *dev;          // device buffer, may be null (= no plan: general kernels only)
  int num_fit, num_spill, max_fit_nodes, max_fit_edges, m, nnz, f, num_edge_global;
  int num_dense;           // the first num_dense fit ranges carry kPlanDense (gt_dense.hip)
  int num_dense_wide = 0;  // ... and the first num_dense_wide of those have more than 128 nodes
  int coords_off = 0;      // int32 offset of the packed edge coordinates of the dense ranges (plan.hip)
  const unsigned short *coords() const { return reinterpret_cast<const unsigned short *>(dev + coords_off); }
  // the edge bitmaps behind the coordinates: mask[8 m], maskT[8 m] (see Csr)
  size_t mask_off() const { return plan_mask_off((size_t)coords_off, nnz); }
  const unsigned *mask() const { return reinterpret_cast<const unsigned *>(dev + mask_off()); }
  const unsigned *maskT() const { return mask() + 8 * (size_t)m; }
  // ... and behind the bitmaps the coordinates once more, in RANK order: position row_ptr[i] + k holds (row, column) of the
  // k-th edge of row i by increasing column -- the order the bitmap-driven forward writes attention values in
  // (gt_dense_fwd_ranked_kernel) and its backward therefore reads them in
  size_t ranked_off() const { return mask_off() + 2 * (size_t)kPlanMaskWords * (size_t)m; }
  const unsigned short *coords_ranked() const { return reinterpret_cast<const unsigned short *>(dev + ranked_off()); }
  const int *fit() const { return dev + kPlanHeader; }
  const int *spill() const { return dev + kPlanHeader + 2 * (size_t)m; }
};

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Pick the lane layout for feature width f.  vec4 requires f % 4 == 0 and 16-byte aligned bases.
// fn is a generic lambda taking a FeatCfg value; returns its int result, or kErrUnsupported.
template <class F>
int dispatch_vec4(int f, F &&fn) {
  // Up to f = 256 a row is held by at most 16 lanes (one DPP row): the SDDMM reduction stays in
  // VALU-rate DPP ops and a wave touches >= 4 rows per gather instruction.
  if (f <= 16) return fn(FeatCfg<4, 4, 1>{});
  if (f <= 32) return fn(FeatCfg<8, 4, 1>{});
  if (f <= 64) return fn(FeatCfg<16, 4, 1>{});
  if (f <= 128) return fn(FeatCfg<16, 4, 2>{});
  if (f <= 256) return fn(FeatCfg<16, 4, 4>{});
  if (f <= 512) return fn(FeatCfg<32, 4, 4>{});
  if (f <= 1024) return fn(FeatCfg<64, 4, 4>{});
  return kErrUnsupported;
}

template <class F>
int dispatch_cfg(int f, bool vec4, F &&fn) {
  if (vec4) return dispatch_vec4(f, fn);
  if (f <= 8) return fn(FeatCfg<8, 1, 1>{});
  if (f <= 16) return fn(FeatCfg<16, 1, 1>{});
  if (f <= 32) return fn(FeatCfg<32, 1, 1>{});
  if (f <= 64) return fn(FeatCfg<64, 1, 1>{});
  if (f <= 128) return fn(FeatCfg<64, 1, 2>{});
  if (f <= 256) return fn(FeatCfg<64, 1, 4>{});
  return kErrUnsupported;
}

inline int launch_status() { return static_cast<int>(hipGetLastError()); }
// hipFuncSetAttribute(fn, MaxDynamicSharedMemorySize, 160 KB), issued once per (device, kernel) and remembered (capi.hip)
int set_max_lds_cached_ptr(const void *fn);

// ---- launchers implemented in the kernel translation units -----------------------------------
int launch_gt_hyper_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *attn_edge,
                        float *out, const int *chunks, int nchunks, hipStream_t s);
bool block_width_ok(int f);
int launch_gt_block_fwd(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                        float *attn_edge, float *edge_ws, float *out, hipStream_t s);
// matrix-core kernels over the first p.num_dense fit ranges (GT: unit edge values only); DFGNN_DENSE=0 in the
// environment (diagnostic switch, read once) keeps every range on the edge-walking kernels
bool dense_enabled();
int launch_gt_dense_fwd(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                        float *attn_edge, float *out, hipStream_t s);
int launch_gt_dense_bwd(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                        const float *attn_edge, const float *grad_out, float *dQ, float *dK, float *dV,
                        hipStream_t s, bool ranked = false);
// ranges a backward keeps in plan position (the largest); it takes the others in reverse plan order (gt_dense.hip says why)
int bwd_reverse_keep(int num_dense);
// the statistics-saving training pair (gt_dense_stats.hip): no attn_edge, row statistics [m, h] instead
int launch_gt_dense_fwd_stats(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V, float *out,
                              float *stat_max, float *stat_sum, hipStream_t s);
int launch_gt_dense_bwd_stats(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                              const float *stat_max, const float *stat_sum, const float *grad_out, float *dQ, float *dK,
                              float *dV, hipStream_t s);
// ... with edge values (gt_dense_stats_w.hip): g.wdense = the plan's dense weights (dfgnn_plan_dense_weights); the two
// launchers above hand over to these when g.wdense is set.  stat_max / stat_sum of the forward may be null (inference).
int launch_gt_dense_fwd_stats_w(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V, float *out,
                                float *stat_max, float *stat_sum, hipStream_t s);
int launch_gt_dense_bwd_stats_w(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                                const float *stat_max, const float *stat_sum, const float *grad_out, float *dQ, float *dK,
                                float *dV, hipStream_t s);
// the attn_edge pair with the values in RANK order (gt_dense_stats_w.hip; one head): the forward is bitmap-driven, the
// backward is launch_gt_dense_bwd with ranked = true (the plan's rank-ordered coordinates)
int launch_gt_dense_fwd_ranked(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                               float *attn_ranked, float *out, hipStream_t s);
// edge_max / edge_sum (nullable): row statistics for the GAT training pair
int launch_gat_dense_fwd(const Csr &g, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, float *out, hipStream_t s, float *edge_max = nullptr,
                         float *edge_sum = nullptr, const float *edge_mask = nullptr, float attn_drop = 0.f);
int launch_gat_dense_bwd(const Csr &g, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, const float *edge_max, const float *edge_sum, const float *grad_out,
                         float *grad_feat, float *grad_row, float *grad_col, hipStream_t s,
                         const float *edge_mask = nullptr, float attn_drop = 0.f);
int launch_gt_tiling_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *out,
                         hipStream_t s);
int launch_gt_sddmm(const Csr &g, const float *Q, const float *K, float *logits, hipStream_t s);
// node-parallel CSR baselines (csr_fwd.hip): 'csr' (row logits in LDS) / 'csr_gm' (in global memory), GAT 'hyper_recompute'
int launch_gt_csr_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *logits, float *out,
                      bool use_lds, hipStream_t s);
int launch_gat_recompute_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                             float *out, hipStream_t s);
int launch_softmax_spmm(const Csr &g, const float *logits, const float *X, float *out, bool use_lds,
                        hipStream_t s);
int launch_gt_bwd_rows(const Csr &g, const float *K, const float *V, const float *attn_edge,
                       const float *grad_out, float *grad_edge, float *dQ, const int *chunks, int nchunks,
                       hipStream_t s);
int launch_gt_bwd_cols(const Csr &g, const int *col_ptr, const int *row_ind, const int *val_idx, const float *Q,
                       const float *attn_edge, const float *grad_edge, const float *grad_out, float *dK,
                       float *dV, const int *chunks, int nchunks, hipStream_t s);
int launch_gt_block_bwd(const Csr &g, const Plan &p, const int *col_ptr, const int *row_ind, const int *val_idx,
                        const float *Q, const float *K, const float *V, const float *attn_edge,
                        const float *grad_out, float *edge_ws, float *dQ, float *dK, float *dV, hipStream_t s);
int launch_gt_lowdeg_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *attn_edge, float *out,
                         hipStream_t s);
int launch_gt_lowdeg_bwd(const Csr &g, const int *col_ptr, const int *row_ind, const int *val_idx, const float *Q,
                         const float *K, const float *V, const float *attn_edge, const float *grad_out,
                         float *grad_edge, float *dQ, float *dK, float *dV, hipStream_t s);
// graphs with fewer than kBlockMinAvgDegree edges per row on average take the row-per-lane-group kernels
inline bool low_degree(int m, int nnz) { return (long)nnz < (long)kBlockMinAvgDegree * m; }
int launch_gat_hyper_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope,
                         const float *X, float *out, const int *chunks, int nchunks, hipStream_t s);
int launch_gat_block_fwd(const Csr &g, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, float *edge_ws, float *out, hipStream_t s);
int launch_gat_tiling_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope,
                          const float *X, float *out, hipStream_t s);
int launch_gat_sddmm(const Csr &g, const float *attn_row, const float *attn_col, float slope, float *logits,
                     hipStream_t s);

int launch_gat_attn_scores(int m, int h, int f, const float *a_l, const float *a_r, const float *X, float *attn_row,
                           float *attn_col, hipStream_t s);
// GAT training pair (gat_train.hip).  rest != NULL: only the ranges of that plan which the matrix-core kernels do not
// serve (its non-dense fit ranges and its spill chunks); NULL: the whole graph.
int launch_gat_train_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                         const float *edge_mask, float attn_drop, float *edge_max, float *edge_sum, float *out,
                         hipStream_t s, const Plan *rest = nullptr);
int launch_gat_bwd_rows(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                        const float *edge_max, const float *edge_sum, const float *edge_mask, float attn_drop,
                        const float *grad_out, float *grad_edge, float *grad_row, hipStream_t s,
                        const Plan *rest = nullptr);
int launch_gat_bwd_cols(const Csr &g, const int *col_ptr, const int *row_ind, const int *permute,
                        const float *attn_row, const float *attn_col, float slope, const float *edge_max,
                        const float *edge_sum, const float *edge_mask, float attn_drop, const float *grad_edge,
                        const float *grad_out, float *grad_feat, float *grad_col, hipStream_t s,
                        const Plan *rest = nullptr);

}  // namespace dfgnn
