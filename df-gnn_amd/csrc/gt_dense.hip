// gt_dense.hip -- matrix-core GT forward and backward for the dense ranges of a block plan.
//
// One workgroup of 8 waves per dense range (single-head; multi-head: per range for the forward and for the backward of
// ranges of <= 128 nodes, which walk the heads -- dfgnn_dense_heads.hpp -- else per (range, head)); wave w owns the
// 16-row strip w (and strip w + 8 of a range with more than 128 nodes).  The sparse structure of a range arrives as
// one uint16 per edge (row << 8 | column inside the range: the plan's `coords`, plan.hip).  See dfgnn_dense.hpp for the
// numerics and the operand layouts.  These kernels replace,
// for such ranges, the same reference kernels as gt_block.hip / gt_block_bwd.hip (fused_gtconv_hyper.cu:228-560,
// fused_gtconv_backward.cu:40-191): the dot products, the softmax and the weighted sums of a whole member graph
// are done as masked dense attention on v_mfma_f32_16x16x32_f16 (fp16 hi / lo operand halves under power-of-two
// scales: fp32-equivalent, see dfgnn_dense.hpp).
//
// Forward:   S^T = K Q^T  ->  masked row softmax in registers  ->  O^T = V^T P^T
//            The mask is a byte map [i][j] -> position of edge (i, j) in row i (0xFF: no edge), built once per
//            range in LDS from the CSR arrays; it also tells where P_ij goes in attn_edge.
// Backward:  P (attn_edge) is scattered into a dense tile (one-tile GT ranges: directly as fp16 hi | lo halves; else fp32,
//            converted in place by the strips); off-edge pairs have P = 0, hence dS = 0: no mask.
//            dV^T = dO^T P first (dO image resident; the strips take their dO rows -- the register operand of the next
//            product -- from it, so dO is read from global memory once), then
//            dP^T = V dO^T ;  t_i = sum_j P_ij dP_ij ;  dS = P o (dP - t)        (registers)
//            dQ^T = K^T dS^T ,  dK^T = Q^T dS                                    (P / dS through an fp16 tile in LDS)
//            A row block of 128 rows keeps dP / P / dS of all (one or two) 128-column blocks in registers, so t_i
//            needs no extra sweep and dQ accumulates in registers; dK / dV of a two-block range are accumulated
//            across the row blocks by the lane that wrote them.
// LDS: one feature image (K then V; dO, V, K, Q in turn; 72 KB) + the byte map (forward) or one 128 x 136-float
// tile (backward: P as fp32, then P and dS as interleaved fp16 hi | lo rows, 68 KB).  The next image's global
// loads are issued one phase ahead into registers, so a phase change costs a barrier and an LDS store.  Every barrier
// is lds_barrier() (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also wait for vmcnt(0), i.e. for the
// image that was just prefetched -- no thread ever reads another thread's global writes in these kernels.
#include <cstdlib>
#include <type_traits>

#define DFGNN_STAMPS_TU  // this translation unit owns the stamp / trace variables of diagnostic builds
#include "dfgnn_dense.hpp"
#include "dfgnn_dense_stamp.hpp"
#ifndef DFGNN_FW128
#define DFGNN_FW128 128
#endif
#ifndef DFGNN_RING160
#define DFGNN_RING160 2  // prefetch distance (image phases) of the 129..160-node backward
#endif
#include "dfgnn_dense_wide.hpp"
#include "dfgnn_dense_lean.hpp"
#include "dfgnn_dense_heads.hpp"
#include "dfgnn_dense_fwd.hpp"
#include "dfgnn_dense_bwd.hpp"
namespace dfgnn {

template <int F, bool WRITE_ATTN>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_fwd_kernel(Csr g, const int *__restrict__ fit,
                                                                     const float *__restrict__ Q,
                                                                     const float *__restrict__ K,
                                                                     const float *__restrict__ V,
                                                                     float *__restrict__ attn_edge,
                                                                     float *__restrict__ out, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  DFGNN_TRACE_IN
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if (g.h == 1) {
    if (n <= kDenseChunkRows)
      dense_fwd_body<F, WRITE_ATTN, 1, kDenseChunkRows, 1>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_edge, out);
    else if (n <= kDenseWideRows)
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseWideRows, 1>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_edge, out);
    else
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseChunkRows, 2>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_edge, out);
  } else {  // every head of the range in this workgroup: edge loads and byte map once
    if constexpr (F == 16 || F == 32 || F == 64) {
      if (dense_heads_ok(F, g.h) && n <= kDenseWideRows) {  // heads in groups of 64 columns (dfgnn_dense_heads.hpp)
#ifdef DFGNN_HEADS_VARIANT  // diagnostic builds: one geometry only (register use of a single body)
        constexpr int hv = DFGNN_HEADS_VARIANT;
#else
        constexpr int hv = -1;
#endif
        if ((hv < 0 && n <= kDenseChunkRows) || hv == 0)
          dense_fwd_heads_body<F, WRITE_ATTN, 1, kDenseChunkRows>(lds, lds_bytes, g, n0, n, e0, ne, Q, K, V, attn_edge, out);
        else if (hv != 0)
          dense_fwd_heads_body<F, WRITE_ATTN, 2, kDenseWideRows>(lds, lds_bytes, g, n0, n, e0, ne, Q, K, V, attn_edge, out);
        DFGNN_TRACE_OUT
        return;
      }
    }
    if (n <= kDenseChunkRows)
      dense_fwd_body<F, WRITE_ATTN, 1, kDenseChunkRows, 1, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_edge, out);
    else if (n <= kDenseWideRows)
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseWideRows, 1, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_edge, out);
    else
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseChunkRows, 2, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_edge, out);
  }
  DFGNN_TRACE_OUT
#ifdef DFGNN_STAMPS
  if (threadIdx.x == 0 && dfgnn_dense_stamps)
    dfgnn_dense_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 15] = ((unsigned long long)n << 32) | (unsigned)ne;
#endif
}

// GAT 'hyper' forward over the dense ranges: replaces fused_gat_hyper_inference{,_vec4}
// (DFGNN/src/fused_gatconv/fused_gatconv_hyper.cu:5-224) for them.
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gat_dense_fwd_kernel(Csr g, const int *__restrict__ fit,
                                                                      const float *__restrict__ attn_row,
                                                                      const float *__restrict__ attn_col, float slope,
                                                                      const float *__restrict__ X,
                                                                      float *__restrict__ out, int lds_bytes,
                                                                      float *__restrict__ edge_max,
                                                                      float *__restrict__ edge_sum, DenseDrop drop) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if (n <= kDenseChunkRows)
    dense_fwd_body<F, false, 1, kDenseChunkRows, 1, true>(lds, lds_bytes, g, n0, n, e0, ne, blockIdx.y, 1, attn_row, attn_col, X,
                                                          nullptr, out, slope, edge_max, edge_sum, drop);
  else if (n <= kDenseWideRows)
    dense_fwd_body<F, false, 2, kDenseWideRows, 1, true>(lds, lds_bytes, g, n0, n, e0, ne, blockIdx.y, 1, attn_row, attn_col, X,
                                                         nullptr, out, slope, edge_max, edge_sum, drop);
  else
    dense_fwd_body<F, false, 2, kDenseChunkRows, 2, true>(lds, lds_bytes, g, n0, n, e0, ne, blockIdx.y, 1, attn_row, attn_col, X,
                                                          nullptr, out, slope, edge_max, edge_sum, drop);
}

// GAT training backward over the dense ranges (no attention dropout): replaces mhspmm_backward_kernel + mhsddmm +
// fused_backward_kernel (DFGNN/src/fused_gatconv/fused_gatconv_kernel.cu:609-865) for them.
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gat_dense_bwd_kernel(
    Csr g, const int *__restrict__ fit, const float *__restrict__ attn_row, const float *__restrict__ attn_col,
    const float *__restrict__ X, const float *__restrict__ dO, float *__restrict__ grad_feat, GatBwdArgs ga) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if (n <= kDenseChunkRows)
    dense_bwd_body<F, kDenseChunkRows, 1, true>(lds, g, n0, n, e0, ne, blockIdx.y, attn_row, attn_col, X, nullptr, dO,
                                                nullptr, nullptr, grad_feat, ga);
  else if (n <= kDenseWideRows)
    dense_bwd_body<F, kDenseWideRows, 1, true>(lds, g, n0, n, e0, ne, blockIdx.y, attn_row, attn_col, X, nullptr, dO,
                                               nullptr, nullptr, grad_feat, ga);
  else
    dense_bwd_body<F, kDenseChunkRows, 2, true>(lds, g, n0, n, e0, ne, blockIdx.y, attn_row, attn_col, X, nullptr, dO,
                                                nullptr, nullptr, grad_feat, ga);
}

template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_bwd_kernel(
    Csr g, const int *__restrict__ fit, const float *__restrict__ Q, const float *__restrict__ K,
    const float *__restrict__ V, const float *__restrict__ attn_edge, const float *__restrict__ dO,
    float *__restrict__ dQ, float *__restrict__ dK, float *__restrict__ dV, int heads_from, int walk, int keep) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  DFGNN_TRACE_IN
  // heads_from < 0: grid (ranges, heads), one workgroup per (range, head).  heads_from >= 0 (multi-head, heads of at
  // most 64 features): a 1-D grid -- the first heads_from ranges (those of more than 128 nodes: they come first in the
  // plan) still get one workgroup per head, every other range ceil(h / walk) workgroups that walk `walk` heads each
  // (dfgnn_dense_heads.hpp).  The 129..160-node ranges stay per head: walked (the body takes NP = 160 for heads of at
  // most 32 features) their workgroups are 8 heads long, 312 of them on 256 CUs, and the kernel waits for the second
  // round -- measured 379 us (8 heads) / 223 us (4 heads) against 387 / 210 us per head.
  // (Workgroups go to the XCDs round-robin by index: the per-head index is range * h + head, so all heads are spread.)
  int range = blockIdx.x, head = blockIdx.y, nwalk = 0;
  if (heads_from >= 0) {
    const int w = blockIdx.x;
    if (w < heads_from * g.h) {
      range = w / g.h;
      head = w - range * g.h;
    } else {
      const int chunks = (g.h + walk - 1) / walk, k = w - heads_from * g.h;
      range = heads_from + k / chunks;
      head = (k % chunks) * walk;
      nwalk = min(walk, g.h - head);
    }
  }
  // keep >= 0 (one head): the first `keep` ranges of the plan (the largest) in place, the others in REVERSE plan order.  The
  // forward walks the plan front to back, so the backward launched right behind it then starts with the ranges whose Q, K,
  // V and attention values the forward touched last -- still in the Infinity Cache -- instead of those it touched first,
  // which the rest of the forward has pushed out (launch_gt_dense_bwd says what that costs and buys).
  if (keep >= 0 && heads_from < 0 && range >= keep) range = (int)gridDim.x - 1 - (range - keep);
  const int n0 = fit[2 * range], n1 = fit[2 * range + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if constexpr (F <= 64) {
    if (nwalk > 0) {  // (n <= 128 by construction)
      dense_bwd_heads_body<F>(lds, g, n0, n, e0, ne, head, nwalk, Q, K, V, attn_edge, dO, dQ, dK, dV);
      DFGNN_TRACE_OUT
      return;
    }
  }
#ifdef DFGNN_BWD_VARIANT  // diagnostic builds: one geometry only (register use / ISA of a single body)
  constexpr int only = DFGNN_BWD_VARIANT;
#else
  constexpr int only = -1;
#endif
  if ((only < 0 && n <= kDenseChunkRows) || only == 0)
#ifdef DFGNN_RING128  // A/B builds: half-width images through the prefetch ring for the <= 128-node ranges as well
    dense_bwd_wide_body<F, kDenseChunkRows, DFGNN_RING128, DFGNN_FW128>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#else
    dense_bwd_body<F, kDenseChunkRows, 1>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#endif
  else if ((only < 0 && n <= kDenseWideRows) || only == 1)
#ifdef DFGNN_OLD_WIDE_BWD  // A/B builds only: two row blocks of <= 80 rows, full-width images
    dense_bwd_body<F, kDenseWideRows, 1>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#else
    dense_bwd_wide_body<F, kDenseWideRows, DFGNN_RING160>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#endif
  else
    dense_bwd_body<F, kDenseChunkRows, 2>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
  DFGNN_TRACE_OUT
#ifdef DFGNN_STAMPS
  if (threadIdx.x == 0 && dfgnn_dense_stamps)
    dfgnn_dense_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 15] = ((unsigned long long)n << 32) | (unsigned)ne;
#endif
}

// =====================================================================================================================
// launchers: the first p.num_dense entries of the plan's fit list
// =====================================================================================================================
bool dense_enabled() {
  static const bool on = [] { const char *e = getenv("DFGNN_DENSE"); return !e || atoi(e) != 0; }();
  return on;
}

template <class Fn>
static int dispatch_dense(int f, Fn &&fn) {
  if (f == 8) return fn(std::integral_constant<int, 8>{});    // f = 8 / 16: zero-padded onto the 32-wide layout
  if (f == 16) return fn(std::integral_constant<int, 16>{});
  if (f == 32) return fn(std::integral_constant<int, 32>{});
  if (f == 64) return fn(std::integral_constant<int, 64>{});
  if (f == 128) return fn(std::integral_constant<int, 128>{});
  return kErrUnsupported;
}

// Heads a workgroup of the multi-head backward walks (ranges of <= 128 nodes): all of them (8 heads of 16: 393 us with
// 8, 397 with 4, 408 with 2 heads per workgroup, 439 per head).  DFGNN_HEADS_WALK in the environment (diagnostic switch,
// read once) overrides it; 0 = one workgroup per (range, head) throughout.
static int heads_walk() {
  static const int w = [] { const char *e = getenv("DFGNN_HEADS_WALK"); return e ? max(0, atoi(e)) : 64; }();
  return w;
}

// DFGNN_LEAN=0 in the environment (diagnostic switch, read once): every dense range on the 512-thread forward
static bool lean_enabled() {
  static const bool on = [] { const char *e = getenv("DFGNN_LEAN"); return !e || atoi(e) != 0; }();
  return on;
}

static int launch_gt_dense_fwd_single(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                                      float *attn_edge, float *out, hipStream_t s) {
  Csr g = g_in;
  g.coords = p.coords();
  const dim3 grid(p.num_dense, 1);
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (attn_edge) {
      if (int rc = set_max_lds_cached(gt_dense_fwd_kernel<F, true>)) return rc;
      gt_dense_fwd_kernel<F, true><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_edge, out, kLdsBytes);
    } else {
      if (int rc = set_max_lds_cached(gt_dense_fwd_kernel<F, false>)) return rc;
      gt_dense_fwd_kernel<F, false><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, nullptr, out, kLdsBytes);
    }
    return launch_status();
  });
}

// The ranges of more than 128 nodes need the 512-thread kernel (one workgroup per CU); a batch without any takes the
// 256-thread kernel, two workgroups per CU (dfgnn_dense_lean.hpp: ~12 % faster on such batches).  A MIXED batch stays on the
// 512-thread kernel as a whole: run as two kernels the classes serialise -- one after the other on the caller's stream the
// second waits for the first one's tail (forward 108 -> 130 us on the headline batch), and forked onto a side stream of the
// library's own (fork / join events) the two grids still ran back to back on this runtime (127 us).
int launch_gt_dense_fwd(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                        float *attn_edge, float *out, hipStream_t s) {
  if (p.num_dense == 0) return 0;
  const bool lean = (g_in.f == 64 || g_in.f == 128) && g_in.h == 1 && p.num_dense_wide == 0 && lean_enabled();
  if (!lean) return launch_gt_dense_fwd_single(g_in, p, Q, K, V, attn_edge, out, s);
  Csr g = g_in;
  g.coords = p.coords();
  const dim3 grid(p.num_dense, 1);
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if constexpr (F == 64 || F == 128) {
      if (attn_edge) {
        if (int rc = set_max_lds_cached(gt_dense_fwd_lean_kernel<F, true>)) return rc;
        gt_dense_fwd_lean_kernel<F, true><<<grid, kLeanThreads, kLeanLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_edge, out);
      } else {
        if (int rc = set_max_lds_cached(gt_dense_fwd_lean_kernel<F, false>)) return rc;
        gt_dense_fwd_lean_kernel<F, false><<<grid, kLeanThreads, kLeanLdsBytes, s>>>(g, p.fit(), Q, K, V, nullptr, out);
      }
    }
    return launch_status();
  });
}

int launch_gat_dense_fwd(const Csr &g_in, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, float *out, hipStream_t s, float *edge_max, float *edge_sum,
                         const float *edge_mask, float attn_drop) {
  Csr g = g_in;
  g.coords = p.coords();
  if (p.num_dense == 0) return 0;
  const dim3 grid(p.num_dense, g.h);
  DenseDrop drop;
  if (edge_mask) drop = DenseDrop{edge_mask, attn_drop, 1.f / (1.f - attn_drop)};
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds_cached(gat_dense_fwd_kernel<F>)) return rc;
    gat_dense_fwd_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), attn_row, attn_col, slope, X, out, kLdsBytes,
                                                                   edge_max, edge_sum, drop);
    return launch_status();
  });
}

int launch_gat_dense_bwd(const Csr &g_in, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, const float *edge_max, const float *edge_sum, const float *grad_out,
                         float *grad_feat, float *grad_row, float *grad_col, hipStream_t s, const float *edge_mask,
                         float attn_drop) {
  Csr g = g_in;
  g.coords = p.coords();
  if (p.num_dense == 0) return 0;
  const dim3 grid(p.num_dense, g.h);
  GatBwdArgs ga{edge_max, edge_sum, slope, grad_row, grad_col, DenseDrop{}};
  if (edge_mask) ga.drop = DenseDrop{edge_mask, attn_drop, 1.f / (1.f - attn_drop)};
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds_cached(gat_dense_bwd_kernel<F>)) return rc;
    gat_dense_bwd_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), attn_row, attn_col, X, grad_out, grad_feat, ga);
    return launch_status();
  });
}

int bwd_reverse_keep(int num_dense) {
  static const int rev = [] { const char *e = getenv("DFGNN_BWD_REVERSE"); return e ? atoi(e) : 8; }();
  return rev > 0 ? num_dense / rev : -1;
}

int launch_gt_dense_bwd(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                        const float *attn_edge, const float *grad_out, float *dQ, float *dK, float *dV,
                        hipStream_t s, bool ranked) {
  Csr g = g_in;
  g.coords = ranked ? p.coords_ranked() : p.coords();  // ranked: attn_edge is in rank order (launch_gt_dense_fwd_ranked)
  if (p.num_dense == 0) return 0;
  // multi-head, heads of at most 64 features: one workgroup per range of <= 128 nodes walks the heads (see the kernel)
  const int walk = (g.h > 1 && g.f <= 64) ? min(heads_walk(), g.h) : 0;
  const int heads_from = walk ? p.num_dense_wide : -1;
  // The order the workgroups take the ranges in (one workgroup per range and head): the plan lists them largest first,
  // which is the order that balances the last round, and the forward walks the list front to back.  Walked the same way the
  // backward would start with the ranges whose Q, K, V and attention values the forward touched FIRST -- gone from the
  // 256 MB Infinity Cache by the time the forward is through its 270 MB -- and the forward after it likewise.  So the
  // backward keeps the largest 1/8 of the ranges in front (the long jobs still start first) and takes the others in
  // REVERSE: it begins where the forward ended and ends where the next forward begins.  Headline step 237 -> 226 us
  // (backward 143 -> 135 us, the forward behind it 94 -> 90 us; 1/8 and 1/10 best, 1/4: 237, everything reversed: 277 us
  // -- the long jobs last); timed back to back with itself the backward does not lose either (138 -> 135 us).
  // DFGNN_BWD_REVERSE=k in the environment (read once): keep the largest 1/k, 0 = plan order.
  const int keep = (g.h == 1) ? bwd_reverse_keep(p.num_dense) : -1;  // (several heads: measured with the statistics pair, no gain)
  const int chunks = walk ? (g.h + walk - 1) / walk : 0;
  const dim3 grid = walk ? dim3(p.num_dense_wide * g.h + (p.num_dense - p.num_dense_wide) * chunks, 1) : dim3(p.num_dense, g.h);
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds_cached(gt_dense_bwd_kernel<F>)) return rc;
    gt_dense_bwd_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_edge, grad_out, dQ, dK, dV,
                                                                  heads_from, walk, keep);
    return launch_status();
  });
}

}  // namespace dfgnn

#if defined(DFGNN_LDS_CHECK)  // ---- `make ldscheck` only: the host side of the bounds-checked LDS addressing -------------
#include <mutex>
#include <vector>
namespace dfgnn {
namespace {
std::mutex &lds_reports_lock() { static std::mutex m; return m; }
std::vector<const void *> &lds_reports() { static std::vector<const void *> v; return v; }
// deliberately out of range through one of the checked helpers (the hardware drops the store): `floats` past the launch's LDS
__global__ void lds_selftest_kernel(int floats, float *out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (threadIdx.x == 0) {
    st32_f4(lds, 0, make_float4(1.f, 2.f, 3.f, 4.f));                  // in range: not counted
    st32_f4(lds, (unsigned)floats, make_float4(5.f, 6.f, 7.f, 8.f));   // [floats, floats + 4): counted once
    out[0] = ld32_f4(lds, 0).y;
  }
}
}  // namespace
void lds_report_register(const void *symbol) {
  std::lock_guard<std::mutex> hold(lds_reports_lock());
  lds_reports().push_back(symbol);
}
}  // namespace dfgnn

// out[0] = violations since the last call summed over the translation units (the counters are cleared), out[1..3] = source
// line (dfgnn_dense*.hpp), end offset and limit of the first violation of the first unit that saw one.  Returns the number
// of translation units compiled with the checks (0 would mean "not a checking build"), negative on a HIP error.
extern "C" int dfgnn_debug_lds_report(unsigned *out) {
  std::lock_guard<std::mutex> hold(dfgnn::lds_reports_lock());
  unsigned total[4] = {0, 0, 0, 0};
  const unsigned zero[4] = {0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  for (const void *sym : dfgnn::lds_reports()) {
    unsigned r[4];
    if (hipMemcpyFromSymbol(r, sym, sizeof(r)) != hipSuccess) return -1;
    if (r[0] && !total[0]) { total[1] = r[1]; total[2] = r[2]; total[3] = r[3]; }
    total[0] += r[0];
    if (hipMemcpyToSymbol(sym, zero, sizeof(zero)) != hipSuccess) return -1;
  }
  for (int k = 0; k < 4; ++k) out[k] = total[k];
  return (int)dfgnn::lds_reports().size();
}
// One store 16 bytes past `lds_bytes` of dynamic LDS through st32_f4: the next report must show exactly one violation
// with limit = lds_bytes.
extern "C" int dfgnn_debug_lds_selftest(int lds_bytes, float *out, void *stream) {
  dfgnn::lds_selftest_kernel<<<1, 64, lds_bytes, static_cast<hipStream_t>(stream)>>>(lds_bytes / 4, out);
  return (int)hipGetLastError();
}
#endif
