// gt_lowdeg.hip -- GT 'hyper' forward / backward for low-degree graphs (molecules, peptides: ~2 edges per row).
//
// With a handful of edges per row, a wave per row (general kernels) or a 1024-thread workgroup per graph
// (resident kernels) is almost all fixed cost.  Here every group of G lanes (one feature row wide) owns one row:
// a wave works on EPW = 64/G rows at once, everything stays in registers (online softmax per edge, no LDS, no
// barriers) and the loops run to each group's own degree under the exec mask.  Selected by the C ABI when
// nnz < kBlockMinAvgDegree * m.  A wave whose EPW rows include one of more than kGroupMaxDegree entries -- the hubs of
// a citation graph, which one lane group would walk serially while the rest of the wave waits -- takes its rows one at
// a time with the wave-per-row routines of dfgnn_rows.hpp instead (a wave-uniform choice: no barrier).  Same math as fused_gt_hyper / fused_backward_kernel / spmm_backward_kernel of
// the reference (fused_gtconv_hyper.cu:31-163, fused_gtconv_backward.cu:40-191).
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

constexpr int kGroupMaxDegree = 24;

// Per-wave LDS scratch (64 weights + 64 column ids) of the wave-per-row fallback.
struct WaveScratch {
  float *sw;
  int *sc;
  int lane, wave;
};
__device__ __forceinline__ WaveScratch wave_scratch() {
  __shared__ __attribute__((aligned(16))) float lds_fb[kWavesPerBlock * kScratchFloatsPerWave];
  const int wave = threadIdx.x / kWave;
  float *sw = lds_fb + wave * kScratchFloatsPerWave;
  return WaveScratch{sw, reinterpret_cast<int *>(sw + kWave), (int)(threadIdx.x & (kWave - 1)), wave};
}

template <class C, bool WRITE_ATTN>
__global__ __launch_bounds__(kBlock) void gt_rowgroup_fwd_kernel(Csr g, const float *__restrict__ Q,
                                                                 const float *__restrict__ K,
                                                                 const float *__restrict__ V,
                                                                 float *__restrict__ attn_edge,
                                                                 float *__restrict__ out) {
  constexpr int G = C::G;
  const int head = blockIdx.y, f = g.f;
  const size_t hf = (size_t)g.h * f, hoff = (size_t)head * f;
  const int gl = threadIdx.x % G;
  float *attn_h = WRITE_ATTN ? attn_edge + (size_t)head * g.nnz : nullptr;
  constexpr int R = kBlock / G;  // rows per block
  const WaveScratch ws = wave_scratch();
  const int lane = ws.lane, wave = ws.wave;
  float *sw = ws.sw;
  int *sc = ws.sc;
  for (int b0 = blockIdx.x * R; b0 < g.m; b0 += gridDim.x * R) {
    const int r = b0 + threadIdx.x / G;
    const int lb = r < g.m ? g.row_ptr[r] : 0, deg = r < g.m ? g.row_ptr[r + 1] - lb : 0;
    if (__any(deg > kGroupMaxDegree)) {
      for (int rr = b0 + wave * C::EPW; rr < min(g.m, b0 + (wave + 1) * C::EPW); ++rr) {
        const int lbw = g.row_ptr[rr], degw = g.row_ptr[rr + 1] - lbw;
        Frag<C> qw;
        frag_load<C>(qw, Q + (size_t)rr * hf + hoff, f, gl);
        gt_row_online<C, WRITE_ATTN>(lbw, degw, g.col_ind, g.val, qw, K + hoff, V + hoff, hf, f, sw, sc,
                                     out + (size_t)rr * hf + hoff, WRITE_ATTN ? attn_h + lbw : nullptr, lane);
      }
      continue;
    }
    if (r >= g.m) continue;
    Frag<C> q, acc;
    frag_load<C>(q, Q + (size_t)r * hf + hoff, f, gl);
    frag_zero<C>(acc);
    float m_run = -INFINITY, l_run = 0.f;
    for (int e = 0; e < deg; ++e) {
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
    const float inv = (l_run != 0.f) ? 1.f / l_run : 0.f;
    frag_store_scaled<C>(acc, inv, out + (size_t)r * hf + hoff, f, gl);
    if constexpr (WRITE_ATTN) {
      // lane 0 of the group wrote the raw logits; the same lane turns them into probabilities
      if (gl == 0)
        for (int e = 0; e < deg; ++e) {
          const float s = attn_h[lb + e];
          attn_h[lb + e] = (s == -INFINITY) ? 0.f : fast_exp(s - m_run) * inv;
        }
    }
  }
}

// rows pass: dS_e = P_e (dP_e - sum_row P dP) -> grad_edge; dQ_r = sum_e dS_e val_e K_c
template <class C>
__global__ __launch_bounds__(kBlock) void gt_rowgroup_bwd_rows_kernel(Csr g, const float *__restrict__ K,
                                                                      const float *__restrict__ V,
                                                                      const float *__restrict__ attn_edge,
                                                                      const float *__restrict__ dO,
                                                                      float *__restrict__ grad_edge,
                                                                      float *__restrict__ dQ) {
  constexpr int G = C::G;
  const int head = blockIdx.y, f = g.f;
  const size_t hf = (size_t)g.h * f, hoff = (size_t)head * f;
  const int gl = threadIdx.x % G;
  const float *P_h = attn_edge + (size_t)head * g.nnz;
  float *dS_h = grad_edge + (size_t)head * g.nnz;
  constexpr int R = kBlock / G;  // rows per block
  const WaveScratch ws = wave_scratch();
  const int lane = ws.lane, wave = ws.wave;
  float *sw = ws.sw;
  int *sc = ws.sc;
  for (int b0 = blockIdx.x * R; b0 < g.m; b0 += gridDim.x * R) {
    const int r = b0 + threadIdx.x / G;
    const int lb = r < g.m ? g.row_ptr[r] : 0, deg = r < g.m ? g.row_ptr[r + 1] - lb : 0;
    if (__any(deg > kGroupMaxDegree)) {
      for (int rr = b0 + wave * C::EPW; rr < min(g.m, b0 + (wave + 1) * C::EPW); ++rr) {
        const int lbw = g.row_ptr[rr];
        gt_bwd_row_online<C>(rr, lbw, g.row_ptr[rr + 1] - lbw, g.col_ind, g.val, K + hoff, V + hoff, dO + hoff, P_h, dS_h,
                             hf, f, sw, sc, dQ + hoff, lane);
      }
      continue;
    }
    if (r >= g.m) continue;
    Frag<C> go, acc;
    frag_load<C>(go, dO + (size_t)r * hf + hoff, f, gl);
    frag_zero<C>(acc);
    float t = 0.f;
    for (int e = 0; e < deg; ++e) {
      Frag<C> v;
      frag_load<C>(v, V + (size_t)g.col_ind[lb + e] * hf + hoff, f, gl);
      t = fmaf(P_h[lb + e], lanes_sum<G>(frag_dot<C>(go, v)), t);
    }
    for (int e = 0; e < deg; ++e) {  // dP is recomputed (the V row is an L1 hit) instead of being parked in memory
      const int c = g.col_ind[lb + e];
      Frag<C> v, k;
      frag_load<C>(v, V + (size_t)c * hf + hoff, f, gl);
      frag_load<C>(k, K + (size_t)c * hf + hoff, f, gl);
      const float ds = P_h[lb + e] * (lanes_sum<G>(frag_dot<C>(go, v)) - t);
      if (gl == 0) dS_h[lb + e] = ds;
      frag_fma<C>(acc, g.val ? ds * g.val[lb + e] : ds, k);
    }
    frag_store_scaled<C>(acc, 1.f, dQ + (size_t)r * hf + hoff, f, gl);
  }
}

// cols pass: dV_j = sum P_e dO_i, dK_j = sum dS_e val_e Q_i over the CSC entries of column j
template <class C>
__global__ __launch_bounds__(kBlock) void gt_rowgroup_bwd_cols_kernel(
    Csr g, const int *__restrict__ col_ptr, const int *__restrict__ row_ind, const int *__restrict__ val_idx,
    const float *__restrict__ Q, const float *__restrict__ attn_edge, const float *__restrict__ grad_edge,
    const float *__restrict__ dO, float *__restrict__ dK, float *__restrict__ dV) {
  constexpr int G = C::G;
  const int head = blockIdx.y, f = g.f;
  const size_t hf = (size_t)g.h * f, hoff = (size_t)head * f;
  const int gl = threadIdx.x % G;
  const float *P_h = attn_edge + (size_t)head * g.nnz, *dS_h = grad_edge + (size_t)head * g.nnz;
  constexpr int R = kBlock / G;  // rows per block
  const WaveScratch ws = wave_scratch();
  const int lane = ws.lane, wave = ws.wave;
  for (int b0 = blockIdx.x * R; b0 < g.m; b0 += gridDim.x * R) {
    const int j = b0 + threadIdx.x / G;
    const int lb = j < g.m ? col_ptr[j] : 0, n = j < g.m ? col_ptr[j + 1] - lb : 0;
    if (__any(n > kGroupMaxDegree)) {
      for (int jj = b0 + wave * C::EPW; jj < min(g.m, b0 + (wave + 1) * C::EPW); ++jj) {
        const int lbw = col_ptr[jj];
        gt_bwd_col_wave<C>(jj, lbw, col_ptr[jj + 1] - lbw, row_ind, val_idx, g.val, Q + hoff, dO + hoff, P_h, dS_h, hf, f,
                           dK + hoff, dV + hoff, lane);
      }
      continue;
    }
    if (j >= g.m) continue;
    Frag<C> aK, aV;
    frag_zero<C>(aK);
    frag_zero<C>(aV);
    for (int t = 0; t < n; ++t) {
      const int e = val_idx[lb + t], i = row_ind[lb + t];
      Frag<C> gi, qi;
      frag_load<C>(gi, dO + (size_t)i * hf + hoff, f, gl);
      frag_load<C>(qi, Q + (size_t)i * hf + hoff, f, gl);
      frag_fma<C>(aV, P_h[e], gi);
      frag_fma<C>(aK, g.val ? dS_h[e] * g.val[e] : dS_h[e], qi);
    }
    frag_store_scaled<C>(aK, 1.f, dK + (size_t)j * hf + hoff, f, gl);
    frag_store_scaled<C>(aV, 1.f, dV + (size_t)j * hf + hoff, f, gl);
  }
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
