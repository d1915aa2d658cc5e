// gt_lowdeg.hip -- GT 'hyper' forward / backward for low-degree graphs (molecules, peptides: ~2 edges per row).
//
// With a handful of edges per row, a wave per row (general kernels) or a 1024-thread workgroup per graph
// (resident kernels) is almost all fixed cost.  Here every group of G lanes (one feature row wide) owns one row:
// a wave works on EPW = 64/G rows at once, everything stays in registers (online softmax per edge, no LDS, no
// barriers) and the loops run to each group's own degree under the exec mask.  Selected by the C ABI when
// nnz < kBlockMinAvgDegree * m.  Rows of more than kGroupMaxDegree entries (the hubs of a citation graph) are taken by
// all groups of their wave together (group_row_loop below).  Same math as fused_gt_hyper / fused_backward_kernel / spmm_backward_kernel of
// the reference (fused_gtconv_hyper.cu:31-163, fused_gtconv_backward.cu:40-191).
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

constexpr int kGroupMaxDegree = 24;

// Sum / max over the EPW lane groups of a wave (every group ends up with the result).
template <class C>
__device__ __forceinline__ float groups_sum(float v) {
#pragma unroll
  for (int o = C::G; o < kWave; o <<= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// One row of the forward by a lane group -- or, COOP, by all EPW groups of the wave together: group gid takes the
// edges gid, gid + EPW, ... with its own online-softmax state, and the states are merged at the end.
template <class C, bool WRITE_ATTN, bool COOP>
__device__ __forceinline__ void gt_group_fwd_row(const Csr &g, int r, const float *__restrict__ Q,
                                                 const float *__restrict__ K, const float *__restrict__ V,
                                                 float *__restrict__ attn_h, float *__restrict__ out, size_t hf,
                                                 size_t hoff, int gid, int gl) {
  constexpr int G = C::G;
  const int f = g.f;
  const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
  const int e0 = COOP ? gid : 0, es = COOP ? C::EPW : 1;
  Frag<C> q, acc;
  frag_load<C>(q, Q + (size_t)r * hf + hoff, f, gl);
  frag_zero<C>(acc);
  float m_run = -INFINITY, l_run = 0.f;
  for (int e = e0; e < deg; e += es) {
    const int c = g.col_ind[lb + e];
    Frag<C> k, v;
    frag_load<C>(k, K + (size_t)c * hf + hoff, f, gl);
    frag_load<C>(v, V + (size_t)c * hf + hoff, f, gl);
    float s = lanes_sum<G>(frag_dot<C>(q, k));
    if (g.val) s *= g.val[lb + e];
    if constexpr (WRITE_ATTN) {
      if (gl == 0) attn_h[lb + e] = s;  // raw logit, normalised below
    }
    const float m_new = fmaxf(m_run, s);
    const float sc = (m_run == -INFINITY) ? 0.f : fast_exp(m_run - m_new);
    const float p = (s == -INFINITY) ? 0.f : fast_exp(s - m_new);
    l_run = l_run * sc + p;
    frag_scale<C>(acc, sc);
    frag_fma<C>(acc, p, v);
    m_run = m_new;
  }
  if constexpr (COOP) {  // merge the groups' (max, sum, accumulator) states pairwise
#pragma unroll
    for (int o = G; o < kWave; o <<= 1) {
      const float m_o = __shfl_xor(m_run, o, kWave), l_o = __shfl_xor(l_run, o, kWave);
      const float m_new = fmaxf(m_run, m_o);
      const float sa = (m_run == -INFINITY) ? 0.f : fast_exp(m_run - m_new);
      const float sb = (m_o == -INFINITY) ? 0.f : fast_exp(m_o - m_new);
      l_run = l_run * sa + l_o * sb;
#pragma unroll
      for (int ch = 0; ch < C::NCH; ++ch)
#pragma unroll
        for (int k = 0; k < C::VEC; ++k)
          acc.v[ch][k] = acc.v[ch][k] * sa + __shfl_xor(acc.v[ch][k], o, kWave) * sb;
      m_run = m_new;
    }
  }
  const float inv = (l_run != 0.f) ? 1.f / l_run : 0.f;
  if (!COOP || gid == 0) frag_store_scaled<C>(acc, inv, out + (size_t)r * hf + hoff, f, gl);
  if constexpr (WRITE_ATTN) {
    // lane 0 of the group wrote the raw logits of its edges; the same lane turns them into probabilities
    if (gl == 0)
      for (int e = e0; e < deg; e += es) {
        const float s = attn_h[lb + e];
        attn_h[lb + e] = (s == -INFINITY) ? 0.f : fast_exp(s - m_run) * inv;
      }
  }
}

// rows pass: dS_e = P_e (dP_e - sum_row P dP) -> grad_edge; dQ_r = sum_e dS_e val_e K_c
template <class C, bool COOP>
__device__ __forceinline__ void gt_group_bwd_row(const Csr &g, int r, const float *__restrict__ K,
                                                 const float *__restrict__ V, const float *__restrict__ P_h,
                                                 const float *__restrict__ dO, float *__restrict__ dS_h,
                                                 float *__restrict__ dQ, size_t hf, size_t hoff, int gid, int gl) {
  constexpr int G = C::G;
  const int f = g.f;
  const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
  const int e0 = COOP ? gid : 0, es = COOP ? C::EPW : 1;
  Frag<C> go, acc;
  frag_load<C>(go, dO + (size_t)r * hf + hoff, f, gl);
  frag_zero<C>(acc);
  float t = 0.f;
  for (int e = e0; e < deg; e += es) {
    Frag<C> v;
    frag_load<C>(v, V + (size_t)g.col_ind[lb + e] * hf + hoff, f, gl);
    t = fmaf(P_h[lb + e], lanes_sum<G>(frag_dot<C>(go, v)), t);
  }
  if constexpr (COOP) t = groups_sum<C>(t);
  for (int e = e0; e < deg; e += es) {  // dP is recomputed (the V row is an L1 hit) instead of being parked in memory
    const int c = g.col_ind[lb + e];
    Frag<C> v, k;
    frag_load<C>(v, V + (size_t)c * hf + hoff, f, gl);
    frag_load<C>(k, K + (size_t)c * hf + hoff, f, gl);
    const float ds = P_h[lb + e] * (lanes_sum<G>(frag_dot<C>(go, v)) - t);
    if (gl == 0) dS_h[lb + e] = ds;
    frag_fma<C>(acc, g.val ? ds * g.val[lb + e] : ds, k);
  }
  if constexpr (COOP) frag_reduce_groups<C>(acc);
  if (!COOP || gid == 0) frag_store_scaled<C>(acc, 1.f, dQ + (size_t)r * hf + hoff, f, gl);
}

// cols pass: dV_j = sum P_e dO_i, dK_j = sum dS_e val_e Q_i over the CSC entries of column j
template <class C, bool COOP>
__device__ __forceinline__ void gt_group_bwd_col(const Csr &g, int j, const int *__restrict__ col_ptr,
                                                 const int *__restrict__ row_ind, const int *__restrict__ val_idx,
                                                 const float *__restrict__ Q, const float *__restrict__ P_h,
                                                 const float *__restrict__ dS_h, const float *__restrict__ dO,
                                                 float *__restrict__ dK, float *__restrict__ dV, size_t hf, size_t hoff,
                                                 int gid, int gl) {
  const int f = g.f;
  const int lb = col_ptr[j], n = col_ptr[j + 1] - lb;
  Frag<C> aK, aV;
  frag_zero<C>(aK);
  frag_zero<C>(aV);
  for (int t = COOP ? gid : 0; t < n; t += COOP ? C::EPW : 1) {
    const int e = val_idx[lb + t], i = row_ind[lb + t];
    Frag<C> gi, qi;
    frag_load<C>(gi, dO + (size_t)i * hf + hoff, f, gl);
    frag_load<C>(qi, Q + (size_t)i * hf + hoff, f, gl);
    frag_fma<C>(aV, P_h[e], gi);
    frag_fma<C>(aK, g.val ? dS_h[e] * g.val[e] : dS_h[e], qi);
  }
  if constexpr (COOP) {
    frag_reduce_groups<C>(aK);
    frag_reduce_groups<C>(aV);
  }
  if (!COOP || gid == 0) {
    frag_store_scaled<C>(aK, 1.f, dK + (size_t)j * hf + hoff, f, gl);
    frag_store_scaled<C>(aV, 1.f, dV + (size_t)j * hf + hoff, f, gl);
  }
}

// The three kernels share one row loop: a workgroup takes blocks of kBlock / G consecutive rows (columns), a lane
// group per row.  A wave whose EPW rows include one of more than kGroupMaxDegree entries -- a hub of a citation graph,
// which a single lane group would walk serially while the rest of the wave waits -- takes its rows one after the other
// with all its groups on each (COOP).  The choice is wave-uniform (ballot): no barrier, no LDS.
template <class C, class Single, class Coop>
__device__ __forceinline__ void group_row_loop(int m, const int *__restrict__ ptr, Single single, Coop coop) {
  constexpr int G = C::G, R = kBlock / G;
  const int gid = (threadIdx.x & (kWave - 1)) / G, gl = threadIdx.x % G;
  const int wave = threadIdx.x / kWave;
  for (int b0 = blockIdx.x * R; b0 < m; b0 += gridDim.x * R) {
    const int r = b0 + threadIdx.x / G;
    const int deg = r < m ? ptr[r + 1] - ptr[r] : 0;
    if (__any(deg > kGroupMaxDegree)) {
      for (int rr = b0 + wave * C::EPW; rr < min(m, b0 + (wave + 1) * C::EPW); ++rr) coop(rr, gid, gl);
    } else if (r < m) {
      single(r, gid, gl);
    }
  }
}

template <class C, bool WRITE_ATTN>
__global__ __launch_bounds__(kBlock) void gt_rowgroup_fwd_kernel(Csr g, const float *__restrict__ Q,
                                                                 const float *__restrict__ K,
                                                                 const float *__restrict__ V,
                                                                 float *__restrict__ attn_edge,
                                                                 float *__restrict__ out) {
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f, hoff = (size_t)head * g.f;
  float *attn_h = WRITE_ATTN ? attn_edge + (size_t)head * g.nnz : nullptr;
  group_row_loop<C>(
      g.m, g.row_ptr,
      [&](int r, int gid, int gl) { gt_group_fwd_row<C, WRITE_ATTN, false>(g, r, Q, K, V, attn_h, out, hf, hoff, gid, gl); },
      [&](int r, int gid, int gl) { gt_group_fwd_row<C, WRITE_ATTN, true>(g, r, Q, K, V, attn_h, out, hf, hoff, gid, gl); });
}

template <class C>
__global__ __launch_bounds__(kBlock) void gt_rowgroup_bwd_rows_kernel(Csr g, const float *__restrict__ K,
                                                                      const float *__restrict__ V,
                                                                      const float *__restrict__ attn_edge,
                                                                      const float *__restrict__ dO,
                                                                      float *__restrict__ grad_edge,
                                                                      float *__restrict__ dQ) {
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f, hoff = (size_t)head * g.f;
  const float *P_h = attn_edge + (size_t)head * g.nnz;
  float *dS_h = grad_edge + (size_t)head * g.nnz;
  group_row_loop<C>(
      g.m, g.row_ptr,
      [&](int r, int gid, int gl) { gt_group_bwd_row<C, false>(g, r, K, V, P_h, dO, dS_h, dQ, hf, hoff, gid, gl); },
      [&](int r, int gid, int gl) { gt_group_bwd_row<C, true>(g, r, K, V, P_h, dO, dS_h, dQ, hf, hoff, gid, gl); });
}

template <class C>
__global__ __launch_bounds__(kBlock) void gt_rowgroup_bwd_cols_kernel(
    Csr g, const int *__restrict__ col_ptr, const int *__restrict__ row_ind, const int *__restrict__ val_idx,
    const float *__restrict__ Q, const float *__restrict__ attn_edge, const float *__restrict__ grad_edge,
    const float *__restrict__ dO, float *__restrict__ dK, float *__restrict__ dV) {
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f, hoff = (size_t)head * g.f;
  const float *P_h = attn_edge + (size_t)head * g.nnz, *dS_h = grad_edge + (size_t)head * g.nnz;
  group_row_loop<C>(
      g.m, col_ptr,
      [&](int j, int gid, int gl) {
        gt_group_bwd_col<C, false>(g, j, col_ptr, row_ind, val_idx, Q, P_h, dS_h, dO, dK, dV, hf, hoff, gid, gl);
      },
      [&](int j, int gid, int gl) {
        gt_group_bwd_col<C, true>(g, j, col_ptr, row_ind, val_idx, Q, P_h, dS_h, dO, dK, dV, hf, hoff, gid, gl);
      });
}

static dim3 rowgroup_grid(const Csr &g, int G) {
  const long groups_per_block = kBlock / G;
  long blocks = ((long)g.m + groups_per_block - 1) / groups_per_block;
  if (blocks > 16384) blocks = 16384;
  return dim3((unsigned)(blocks < 1 ? 1 : blocks), g.h);
}

int launch_gt_lowdeg_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *attn_edge, float *out,
                         hipStream_t s) {
  const bool v4 = (g.f % 4 == 0) && aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    const dim3 grid = rowgroup_grid(g, C::G);
    if (attn_edge)
      gt_rowgroup_fwd_kernel<C, true><<<grid, kBlock, 0, s>>>(g, Q, K, V, attn_edge, out);
    else
      gt_rowgroup_fwd_kernel<C, false><<<grid, kBlock, 0, s>>>(g, Q, K, V, nullptr, out);
    return launch_status();
  });
}

int launch_gt_lowdeg_bwd(const Csr &g, const int *col_ptr, const int *row_ind, const int *val_idx, const float *Q,
                         const float *K, const float *V, const float *attn_edge, const float *grad_out,
                         float *grad_edge, float *dQ, float *dK, float *dV, hipStream_t s) {
  const bool v4 = (g.f % 4 == 0) && aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(grad_out) &&
                  aligned16(dQ) && aligned16(dK) && aligned16(dV);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    const dim3 grid = rowgroup_grid(g, C::G);
    gt_rowgroup_bwd_rows_kernel<C><<<grid, kBlock, 0, s>>>(g, K, V, attn_edge, grad_out, grad_edge, dQ);
    if (int rc = launch_status()) return rc;
    gt_rowgroup_bwd_cols_kernel<C><<<grid, kBlock, 0, s>>>(g, col_ptr, row_ind, val_idx, Q, attn_edge, grad_edge,
                                                           grad_out, dK, dV);
    return launch_status();
  });
}

}  // namespace dfgnn
