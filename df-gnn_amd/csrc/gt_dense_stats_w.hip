// gt_dense_stats_w.hip -- the statistics-saving GT pair (gt_dense_stats.hip) for WEIGHTED edges: logit = (Q_i . K_j) val_ij,
// the reference's `attn * val` (fused_gtconv_hyper.cu:88-90), on the matrix cores.
//
// The dense kernels know WHERE the edges are from the plan's bitmaps; what an edge value needs in addition is a place a
// lane can fetch it from by (row, column) -- the CSR position of an edge is exactly what these kernels do not have.  So
// the values are laid out once per (plan, val) in dense form, W[256 i + c] = val of the edge from node i to the c-th node
// of its range (plan.hip: dfgnn_plan_dense_weights; 1 KB per node, of which a row touches its range's width), and a lane
// reads the 4 values of a 16-column tile with one 16-byte load where the unit-value kernels read nothing:
//   forward :  x_ij = S_ij W_ij on the bitmap's edges, then the softmax, statistics and P V product unchanged
//   backward:  P recomputed from S W; dS_ij = P_ij (dP_ij - t_i) W_ij goes into the dQ and dK products (d x / d S = W)
// One instance per shape: the range classes of the unit-value pair (<= 128, 129-160, > 160 nodes) per (range, head) --
// the all-heads-in-one-workgroup bodies are not instantiated for weights.  Inference with weights runs the forward
// without statistics (null pointers).
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
#include "dfgnn_dense_fwd.hpp"
#include "dfgnn_dense_bwd.hpp"
#include "dfgnn_dense_bwd_rc2.hpp"

namespace dfgnn {

// grid (dense ranges, heads)
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_fwd_stats_w_kernel(Csr g, const int *__restrict__ fit,
                                                                             const float *__restrict__ Q,
                                                                             const float *__restrict__ K,
                                                                             const float *__restrict__ V,
                                                                             float *__restrict__ out,
                                                                             float *__restrict__ stat_max,
                                                                             float *__restrict__ stat_sum, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, head = blockIdx.y;
  if (n <= kDenseChunkRows)
    dense_fwd_body<F, false, 1, kDenseChunkRows, 1, false, false, true, true>(lds, lds_bytes, g, n0, n, 0, 0, head, 1, Q, K, V,
                                                                              nullptr, out, 0.f, stat_max, stat_sum);
  else if (n <= kDenseWideRows)
    dense_fwd_body<F, false, 2, kDenseWideRows, 1, false, false, true, true>(lds, lds_bytes, g, n0, n, 0, 0, head, 1, Q, K, V,
                                                                             nullptr, out, 0.f, stat_max, stat_sum);
  else
    dense_fwd_body<F, false, 2, kDenseChunkRows, 2, false, false, true, true>(lds, lds_bytes, g, n0, n, 0, 0, head, 1, Q, K, V,
                                                                              nullptr, out, 0.f, stat_max, stat_sum);
}

template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_bwd_stats_w_kernel(
    Csr g, const int *__restrict__ fit, const float *__restrict__ Q, const float *__restrict__ K,
    const float *__restrict__ V, const float *__restrict__ stat_max, const float *__restrict__ stat_sum,
    const float *__restrict__ dO, float *__restrict__ dQ, float *__restrict__ dK, float *__restrict__ dV, int keep) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int range = blockIdx.x;
  const int head = blockIdx.y;
  if (keep >= 0 && range >= keep) range = (int)gridDim.x - 1 - (range - keep);  // (launch_gt_dense_bwd, gt_dense.hip)
  const int n0 = fit[2 * range], n1 = fit[2 * range + 1] & kPlanRangeMask;
  const int n = n1 - n0;
  GatBwdArgs st{};
  st.edge_max = stat_max;
  st.edge_sum = stat_sum;
  if (n <= kDenseChunkRows)
    dense_bwd_rc2_body<F, true>(lds, g, n0, n, head, Q, K, V, stat_max, stat_sum, dO, dQ, dK, dV);
  else if (n <= kDenseWideRows)
    dense_bwd_wide_body<F, kDenseWideRows, DFGNN_RING160, 64, true, true>(lds, g, n0, n, 0, 0, head, Q, K, V, nullptr, dO, dQ, dK,
                                                                         dV, stat_max, stat_sum);
  else
    dense_bwd_body<F, kDenseChunkRows, 2, false, true, true>(lds, g, n0, n, 0, 0, head, Q, K, V, nullptr, dO, dQ, dK, dV, st);
}

// ---- the "ranked" training forward: attn_edge written, but in RANK order -------------------------------------
// The attn_edge pair's forward spends a fifth of its time on the sparse structure: 2 B of coordinates per edge, a byte map
// (cleared, scattered into, read back) to find an edge's CSR position.  What the autograd pair needs is only that forward
// and backward agree on an order -- so this forward takes the edges from the bitmaps like the statistics forward and
// writes the value of row i's k-th edge BY INCREASING COLUMN to row_ptr[i] + k (the slot of a pair = the set bits before
// it), and gt_dense_bwd_kernel reads the values back with the plan's rank-ordered coordinates instead of the CSR-ordered
// ones: the backward is the same kernel at the same cost.  gt_hyper_forward -> [out, attn_edge] for direct callers keeps
// the CSR order (gt_dense_fwd_kernel).
#ifndef DFGNN_RANKED_WRITE
#define DFGNN_RANKED_WRITE true  // (diagnostic builds: false = the same kernel without the attention values, for phase stamps)
#endif
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_fwd_ranked_kernel(Csr g, const int *__restrict__ fit,
                                                                            const float *__restrict__ Q,
                                                                            const float *__restrict__ K,
                                                                            const float *__restrict__ V,
                                                                            float *__restrict__ attn_ranked,
                                                                            float *__restrict__ out, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  // (the three classes as non-inlined functions -- each then keeps its own register allocation, 170 / 192 / 236 VGPRs and no
  // spills instead of 256 and 19 for the merged body -- ran 84 -> 102 us: behind a call the LDS pointers are generic and
  // every ds_ access becomes a flat one)
  if (n <= kDenseChunkRows)
    dense_fwd_body<F, DFGNN_RANKED_WRITE, 1, kDenseChunkRows, 1, false, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_ranked, out);
  else if (n <= kDenseWideRows)
    dense_fwd_body<F, DFGNN_RANKED_WRITE, 2, kDenseWideRows, 1, false, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_ranked, out);
#ifndef DFGNN_RANKED_NO_TWO_CHUNK  // (diagnostic builds: what the rare > 160-node class costs the others in registers)
  else
    dense_fwd_body<F, DFGNN_RANKED_WRITE, 2, kDenseChunkRows, 2, false, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_ranked, out);
#endif
}

// ... several heads: every head of the range in this workgroup, like gt_dense_fwd_kernel (a kernel of its own: its widest
// body must not set the register budget of the one-head kernel above)
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_fwd_ranked_heads_kernel(Csr g, const int *__restrict__ fit,
                                                                                  const float *__restrict__ Q,
                                                                                  const float *__restrict__ K,
                                                                                  const float *__restrict__ V,
                                                                                  float *__restrict__ attn_ranked,
                                                                                  float *__restrict__ out, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if constexpr (F == 16 || F == 32 || F == 64) {
    if (dense_heads_ok(F, g.h) && n <= kDenseWideRows) {  // heads in groups of 64 columns (dfgnn_dense_heads.hpp)
      if (n <= kDenseChunkRows)
        dense_fwd_heads_body<F, true, 1, kDenseChunkRows, true>(lds, lds_bytes, g, n0, n, e0, ne, Q, K, V, attn_ranked, out);
      else
        dense_fwd_heads_body<F, true, 2, kDenseWideRows, true>(lds, lds_bytes, g, n0, n, e0, ne, Q, K, V, attn_ranked, out);
      return;
    }
  }
  if (n <= kDenseChunkRows)
    dense_fwd_body<F, true, 1, kDenseChunkRows, 1, false, true, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_ranked, out);
  else if (n <= kDenseWideRows)
    dense_fwd_body<F, true, 2, kDenseWideRows, 1, false, true, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_ranked, out);
  else
    dense_fwd_body<F, true, 2, kDenseChunkRows, 2, false, true, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_ranked, out);
}

template <class Fn>
static int dispatch_dense_w(int f, Fn &&fn) {
  if (f == 8) return fn(std::integral_constant<int, 8>{});  // f = 8 / 16: zero-padded onto the 32-wide layout
  if (f == 16) return fn(std::integral_constant<int, 16>{});
  if (f == 32) return fn(std::integral_constant<int, 32>{});
  if (f == 64) return fn(std::integral_constant<int, 64>{});
  if (f == 128) return fn(std::integral_constant<int, 128>{});
  return kErrUnsupported;
}

// g.wdense, g.mask, g.maskT set by the caller (launch_gt_dense_fwd_stats / launch_gt_dense_bwd_stats)
int launch_gt_dense_fwd_stats_w(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V, float *out,
                                float *stat_max, float *stat_sum, hipStream_t s) {
  const dim3 grid(p.num_dense, g.h);
  return dispatch_dense_w(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds_cached(gt_dense_fwd_stats_w_kernel<F>)) return rc;
    gt_dense_fwd_stats_w_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, out, stat_max, stat_sum, kLdsBytes);
    return launch_status();
  });
}

int launch_gt_dense_bwd_stats_w(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                                const float *stat_max, const float *stat_sum, const float *grad_out, float *dQ, float *dK,
                                float *dV, hipStream_t s) {
  const dim3 grid(p.num_dense, g.h);
  return dispatch_dense_w(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds_cached(gt_dense_bwd_stats_w_kernel<F>)) return rc;
    gt_dense_bwd_stats_w_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, stat_max, stat_sum, grad_out, dQ,
                                                                          dK, dV, g.h == 1 ? bwd_reverse_keep(p.num_dense) : -1);  // (several heads: measured, no gain)
    return launch_status();
  });
}

static bool ranked_lean_enabled() {  // DFGNN_LEAN=0 (diagnostic switch, read once): every dense range on the 512-thread forward
  static const bool on = [] { const char *e = getenv("DFGNN_LEAN"); return !e || atoi(e) != 0; }();
  return on;
}

int launch_gt_dense_fwd_ranked(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                               float *attn_ranked, float *out, hipStream_t s) {
  if (p.num_dense == 0) return 0;
  Csr g = g_in;
  g.mask = p.mask();
  const dim3 grid(p.num_dense, 1);
  // a one-head batch without ranges of more than 128 nodes: the 256-thread forward, two workgroups per CU
  const bool lean = (g.f == 64 || g.f == 128) && g.h == 1 && p.num_dense_wide == 0 && ranked_lean_enabled();
  return dispatch_dense_w(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if constexpr (F == 64 || F == 128) {
      if (lean) {
        if (int rc = set_max_lds_cached(gt_dense_fwd_lean_kernel<F, true, true>)) return rc;
        gt_dense_fwd_lean_kernel<F, true, true><<<grid, kLeanThreads, kLeanLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_ranked, out, nullptr,
                                                                                         nullptr);
        return launch_status();
      }
    }
    if (g.h == 1) {
      if (int rc = set_max_lds_cached(gt_dense_fwd_ranked_kernel<F>)) return rc;
      gt_dense_fwd_ranked_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_ranked, out, kLdsBytes);
    } else {
      if (int rc = set_max_lds_cached(gt_dense_fwd_ranked_heads_kernel<F>)) return rc;
      gt_dense_fwd_ranked_heads_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_ranked, out, kLdsBytes);
    }
    return launch_status();
  });
}

}  // namespace dfgnn
