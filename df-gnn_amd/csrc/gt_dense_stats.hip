// gt_dense_stats.hip -- the STATISTICS-SAVING training pair of the matrix-core GT kernels: forward without attn_edge,
// backward that recomputes P.
//
// The reference's training pair (FusedGTFunction_hyper, DFGNN/operators/fused_gtconv.py:79-158; fused_gtconv_hyper.cu:
// 31-163 -> fused_gtconv_backward.cu:40-191) hands the normalised attention from the forward to the backward through
// attn_edge[h, nnz]: 4 h nnz bytes written, 4 h nnz read back, a scatter into place on either side.  On the matrix cores
// S = Q K^T costs next to nothing (MFMA pipe 15 % busy), so this pair saves TWO FLOATS PER (row, head)
// instead -- the logit maximum m_i and the sum of exponentials l_i, like the GAT training pair -- and the backward
// recomputes P_ij = exp(S_ij - m_i) / l_i.  The edge SET (all these kernels need of the sparse structure: not the order of
// the edges, not their positions) comes from the plan's bitmaps (plan.hip: mask / maskT, 32 bytes per node), fetched by
// each lane for its own rows: no edge list, no row pointers, no byte map, no scatter.
//   forward :  dense_fwd_body / gt_dense_fwd_lean_kernel / dense_fwd_heads_body with STATS
//   backward:  one head      -- <= 128 nodes: dense_bwd_rc2_body ([dO|V], [Q|K], [dO|P]: images staged in pairs, dQ straight
//                               from the dS accumulators); larger: dense_bwd_wide_body / dense_bwd_body with RECOMP (K again)
//              multi-head    -- dense_bwd_heads2_body (ranges of <= 160 nodes, heads of 16 / 32 / 64 features: every head
//                               of a range in one workgroup, no P / dS tile); other shapes per (range, head) as above
// gt_hyper_forward -> [out, attn_edge] (the reference's signature) stays what it was for direct callers; the autograd
// Function takes this pair when the whole batch is served by the matrix-core kernels (dfgnn_gt_stats_applies).
#include <cstdlib>
#include <type_traits>

#include "dfgnn_dense.hpp"
#include "dfgnn_dense_stamp.hpp"
#ifndef DFGNN_RING160
#define DFGNN_RING160 2
#endif
#include "dfgnn_dense_wide.hpp"
#include "dfgnn_dense_lean.hpp"
#include "dfgnn_dense_heads.hpp"
#include "dfgnn_dense_heads2.hpp"
#include "dfgnn_dense_fwd.hpp"
#include "dfgnn_dense_bwd.hpp"
#include "dfgnn_dense_bwd_rc2.hpp"

namespace dfgnn {

template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_fwd_stats_kernel(Csr g, const int *__restrict__ fit,
                                                                           const float *__restrict__ Q,
                                                                           const float *__restrict__ K,
                                                                           const float *__restrict__ V,
                                                                           float *__restrict__ out,
                                                                           float *__restrict__ stat_max,
                                                                           float *__restrict__ stat_sum, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0;
  if (g.h == 1) {
    if (n <= kDenseChunkRows)
      dense_fwd_body<F, false, 1, kDenseChunkRows, 1, false, false, true>(lds, lds_bytes, g, n0, n, 0, 0, 0, 1, Q, K, V, nullptr,
                                                                          out, 0.f, stat_max, stat_sum);
    else if (n <= kDenseWideRows)
      dense_fwd_body<F, false, 2, kDenseWideRows, 1, false, false, true>(lds, lds_bytes, g, n0, n, 0, 0, 0, 1, Q, K, V, nullptr,
                                                                         out, 0.f, stat_max, stat_sum);
    else
      dense_fwd_body<F, false, 2, kDenseChunkRows, 2, false, false, true>(lds, lds_bytes, g, n0, n, 0, 0, 0, 1, Q, K, V, nullptr,
                                                                          out, 0.f, stat_max, stat_sum);
  } else {
    if constexpr (F == 16 || F == 32 || F == 64) {
      if (dense_heads_ok(F, g.h) && n <= kDenseWideRows) {  // heads in groups of 64 columns (dfgnn_dense_heads.hpp)
        if (n <= kDenseChunkRows)
          dense_fwd_heads_body<F, false, 1, kDenseChunkRows, true>(lds, lds_bytes, g, n0, n, 0, 0, Q, K, V, nullptr, out, stat_max,
                                                                   stat_sum);
        else
          dense_fwd_heads_body<F, false, 2, kDenseWideRows, true>(lds, lds_bytes, g, n0, n, 0, 0, Q, K, V, nullptr, out, stat_max,
                                                                  stat_sum);
        return;
      }
    }
    if (n <= kDenseChunkRows)
      dense_fwd_body<F, false, 1, kDenseChunkRows, 1, false, true, true>(lds, lds_bytes, g, n0, n, 0, 0, 0, g.h, Q, K, V, nullptr,
                                                                         out, 0.f, stat_max, stat_sum);
    else if (n <= kDenseWideRows)
      dense_fwd_body<F, false, 2, kDenseWideRows, 1, false, true, true>(lds, lds_bytes, g, n0, n, 0, 0, 0, g.h, Q, K, V, nullptr,
                                                                        out, 0.f, stat_max, stat_sum);
    else
      dense_fwd_body<F, false, 2, kDenseChunkRows, 2, false, true, true>(lds, lds_bytes, g, n0, n, 0, 0, 0, g.h, Q, K, V, nullptr,
                                                                         out, 0.f, stat_max, stat_sum);
  }
}

// grid (dense ranges, heads).  Multi-head with a 64-column group form: the workgroup of head 0 takes every head of a
// range of <= 160 nodes (dense_bwd_heads2_body), the other heads' workgroups of that range leave at once (their indices
// lie behind all the working ones: workgroups go to the XCDs round-robin by index, so the workers stay spread).
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_bwd_stats_kernel(
    Csr g, const int *__restrict__ fit, const float *__restrict__ Q, const float *__restrict__ K,
    const float *__restrict__ V, const float *__restrict__ stat_max, const float *__restrict__ stat_sum,
    const float *__restrict__ dO, float *__restrict__ dQ, float *__restrict__ dK, float *__restrict__ dV, int heads2, int keep) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int range = blockIdx.x;
  const int head = blockIdx.y;
  if (keep >= 0 && range >= keep) range = (int)gridDim.x - 1 - (range - keep);  // (launch_gt_dense_bwd, gt_dense.hip)
  const int n0 = fit[2 * range], n1 = fit[2 * range + 1] & kPlanRangeMask;
  const int n = n1 - n0;
  if constexpr (F == 16 || F == 32 || F == 64) {
    if (heads2 && n <= kDenseWideRows) {  // (heads2: the launcher's choice, uniform over the grid)
      if (head != 0) return;
      if (n <= kDenseChunkRows) dense_bwd_heads2_body<F, kDenseChunkRows>(lds, g, n0, n, Q, K, V, dO, stat_max, stat_sum, dQ, dK, dV);
      else dense_bwd_heads2_body<F, kDenseWideRows>(lds, g, n0, n, Q, K, V, dO, stat_max, stat_sum, dQ, dK, dV);
      return;
    }
  }
  GatBwdArgs st{};
  st.edge_max = stat_max;
  st.edge_sum = stat_sum;
  if (n <= kDenseChunkRows)
    dense_bwd_rc2_body<F>(lds, g, n0, n, head, Q, K, V, stat_max, stat_sum, dO, dQ, dK, dV);
  else if (n <= kDenseWideRows)
    dense_bwd_wide_body<F, kDenseWideRows, DFGNN_RING160, 64, true>(lds, g, n0, n, 0, 0, head, Q, K, V, nullptr, dO, dQ, dK, dV,
                                                                    stat_max, stat_sum);
  else
    dense_bwd_body<F, kDenseChunkRows, 2, false, true>(lds, g, n0, n, 0, 0, head, Q, K, V, nullptr, dO, dQ, dK, dV, st);
}

template <class Fn>
static int dispatch_dense_stats(int f, Fn &&fn) {
  if (f == 8) return fn(std::integral_constant<int, 8>{});  // f = 8 / 16: zero-padded onto the 32-wide layout
  if (f == 16) return fn(std::integral_constant<int, 16>{});
  if (f == 32) return fn(std::integral_constant<int, 32>{});
  if (f == 64) return fn(std::integral_constant<int, 64>{});
  if (f == 128) return fn(std::integral_constant<int, 128>{});
  return kErrUnsupported;
}

// widest head the all-heads-in-one-workgroup backward takes (DFGNN_HEADS2_MAXF in the environment: diagnostic switch)
static int heads2_max_f() {
  static const int v = [] { const char *e = getenv("DFGNN_HEADS2_MAXF"); return e ? atoi(e) : 16; }();
  return v;
}

static bool stats_lean_enabled() {
  static const bool on = [] { const char *e = getenv("DFGNN_LEAN"); return !e || atoi(e) != 0; }();
  return on;
}

int launch_gt_dense_fwd_stats(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V, float *out,
                              float *stat_max, float *stat_sum, hipStream_t s) {
  if (p.num_dense == 0) return 0;
  Csr g = g_in;
  g.mask = p.mask();
  g.maskT = p.maskT();
  if (g.wdense) return launch_gt_dense_fwd_stats_w(g, p, Q, K, V, out, stat_max, stat_sum, s);  // edge values
  const dim3 grid(p.num_dense, 1);
  // a batch without ranges of more than 128 nodes: the 256-thread forward, two workgroups per CU (dfgnn_dense_lean.hpp)
  const bool lean = (g.f == 64 || g.f == 128) && g.h == 1 && p.num_dense_wide == 0 && stats_lean_enabled();
  return dispatch_dense_stats(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if constexpr (F == 64 || F == 128) {
      if (lean) {
        if (int rc = set_max_lds_cached(gt_dense_fwd_lean_kernel<F, false, true>)) return rc;
        gt_dense_fwd_lean_kernel<F, false, true><<<grid, kLeanThreads, kLeanLdsBytes, s>>>(g, p.fit(), Q, K, V, nullptr, out, stat_max,
                                                                                          stat_sum);
        return launch_status();
      }
    }
    if (int rc = set_max_lds_cached(gt_dense_fwd_stats_kernel<F>)) return rc;
    gt_dense_fwd_stats_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, out, stat_max, stat_sum, kLdsBytes);
    return launch_status();
  });
}

int launch_gt_dense_bwd_stats(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                              const float *stat_max, const float *stat_sum, const float *grad_out, float *dQ, float *dK,
                              float *dV, hipStream_t s) {
  if (p.num_dense == 0) return 0;
  Csr g = g_in;
  g.mask = p.mask();
  g.maskT = p.maskT();
  if (g.wdense) return launch_gt_dense_bwd_stats_w(g, p, Q, K, V, stat_max, stat_sum, grad_out, dQ, dK, dV, s);  // edge values
  const dim3 grid(p.num_dense, g.h);
  // multi-head: every head of a range of <= 160 nodes in one workgroup (dense_bwd_heads2_body) for heads of at most
  // kHeads2MaxF features; wider heads run per (range, head) on the single-head bodies
  const int heads2 = (g.h > 1 && dense_heads_ok(g.f, g.h) && g.f <= heads2_max_f()) ? 1 : 0;
  return dispatch_dense_stats(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds_cached(gt_dense_bwd_stats_kernel<F>)) return rc;
    gt_dense_bwd_stats_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, stat_max, stat_sum, grad_out, dQ, dK,
                                                                        dV, heads2, g.h == 1 ? bwd_reverse_keep(p.num_dense) : -1);  // (several heads: measured, no gain)
    return launch_status();
  });
}

}  // namespace dfgnn
