// gt_bwd.hip -- backward of the fused GT convolution for gfx950 (two launches).
//
//   gt_bwd_rows_kernel  CSR pass: dP_e = <dO[i],V[j]>  (edge-parallel, into LDS)
//                                 dS_e = P_e (dP_e - sum_row P dP)  -> grad_edge
//                                 dQ[i] = sum_e dS_e val_e K[j]      (wave per row)
//                       replaces fused_backward_kernel (DFGNN/src/fused_gtconv/fused_gtconv_backward.cu:73-191)
//   gt_bwd_cols_kernel  CSC pass: dV[j] = sum_e P_e dO[i],  dK[j] = sum_e dS_e val_e Q[i]   (wave per column)
//                       replaces spmm_backward_kernel (fused_gtconv_backward.cu:40-70)
//
// Deliberate divergences from the reference: attn_edge / grad_edge are
// indexed with the head offset (the reference's backward is single-head only) and val is applied
// (the reference drops it; identical for its only live input val == 1).
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

template <class C>
__global__ __launch_bounds__(kBlock) void gt_bwd_rows_kernel(Csr g, const float *__restrict__ K,
                                                             const float *__restrict__ V,
                                                             const float *__restrict__ attn_edge,
                                                             const float *__restrict__ dO,
                                                             float *__restrict__ grad_edge,
                                                             float *__restrict__ dQ,
                                                             const int *__restrict__ chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *lw = lds;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + kHyperCap + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);

  const int head = blockIdx.y;
  const int r0 = chunks ? chunks[2 * blockIdx.x] : blockIdx.x * kHyperRows;
  const int r1 = chunks ? chunks[2 * blockIdx.x + 1] : min(g.m, r0 + kHyperRows);
  const size_t hf = (size_t)g.h * g.f;
  const int f = g.f;
  const float *Kh = K + (size_t)head * f, *Vh = V + (size_t)head * f, *dOh = dO + (size_t)head * f;
  const float *P_h = attn_edge + (size_t)head * g.nnz;
  float *dS_h = grad_edge + (size_t)head * g.nnz;
  float *dQh = dQ + (size_t)head * f;
  const int e0 = g.row_ptr[r0], e1 = g.row_ptr[r1];
  const int ne = e1 - e0;
  const int gid = lane / C::G, gl = lane % C::G;

  if (ne <= kHyperCap) {
    constexpr int NG = kBlock / C::G;
    sddmm_range<C>(e0, e1, threadIdx.x / C::G, NG, g.rows, g.col_ind, nullptr, dOh, Vh, hf, f, gl,
                   [&](int e, float s) { lw[e - e0] = s; });
    __syncthreads();
    for (int r = r0 + wave; r < r1; r += kWavesPerBlock) {
      const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
      float *lr = lw + (lb - e0);
      float t = 0.f;
      for (int e = lane; e < deg; e += kWave) t = fmaf(lr[e], P_h[lb + e], t);
      t = lanes_sum<kWave>(t);
      for (int e = lane; e < deg; e += kWave) {
        const float ds = P_h[lb + e] * (lr[e] - t);
        dS_h[lb + e] = ds;
        lr[e] = g.val ? ds * g.val[lb + e] : ds;
      }
      wave_sync();
      Frag<C> acc;
      frag_zero<C>(acc);
      spmm_accum<C>(acc, lr, g.col_ind + lb, deg, Kh, hf, f, gid, gl);
      frag_reduce_groups<C>(acc);
      if (gid == 0) frag_store_scaled<C>(acc, 1.f, dQh + (size_t)r * hf, f, gl);
    }
  } else {
    // heavy rows: 64-edge tiles, dP parked in grad_edge between the two sweeps
    for (int r = r0 + wave; r < r1; r += kWavesPerBlock) {
      const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
      gt_bwd_row_online<C>(r, lb, deg, g.col_ind, g.val, Kh, Vh, dOh, P_h, dS_h, hf, f, sw, sc, dQh, lane);
    }
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gt_bwd_cols_kernel(Csr g, const int *__restrict__ col_ptr,
                                                             const int *__restrict__ row_ind,
                                                             const int *__restrict__ val_idx,
                                                             const float *__restrict__ Q,
                                                             const float *__restrict__ attn_edge,
                                                             const float *__restrict__ grad_edge,
                                                             const float *__restrict__ dO,
                                                             float *__restrict__ dK, float *__restrict__ dV,
                                                             const int *__restrict__ chunks) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f;
  const int f = g.f;
  const float *Qh = Q + (size_t)head * f, *dOh = dO + (size_t)head * f;
  const float *P_h = attn_edge + (size_t)head * g.nnz, *dS_h = grad_edge + (size_t)head * g.nnz;
  // columns of this workgroup: grid-strided over all columns, or (with a block plan) one spill chunk
  const int jbeg = chunks ? chunks[2 * blockIdx.x] + wave : blockIdx.x * kWavesPerBlock + wave;
  const int jend = chunks ? chunks[2 * blockIdx.x + 1] : g.m;
  const int jstep = chunks ? kWavesPerBlock : gridDim.x * kWavesPerBlock;
  for (int j = jbeg; j < jend; j += jstep) {
    const int lb = col_ptr[j], n = col_ptr[j + 1] - lb;
    gt_bwd_col_wave<C>(j, lb, n, row_ind, val_idx, g.val, Qh, dOh, P_h, dS_h, hf, f, dK + (size_t)head * f,
                       dV + (size_t)head * f, lane);
  }
}

static inline bool bwd_vec4(const Csr &g, const float *a, const float *b, const float *c, const float *d,
                            const float *e) {
  return (g.f % 4 == 0) && aligned16(a) && aligned16(b) && aligned16(c) && aligned16(d) && aligned16(e);
}

int launch_gt_bwd_rows(const Csr &g, const float *K, const float *V, const float *attn_edge,
                       const float *grad_out, float *grad_edge, float *dQ, const int *chunks, int nchunks,
                       hipStream_t s) {
  const dim3 grid(chunks ? nchunks : (g.m + kHyperRows - 1) / kHyperRows, g.h);
  if (grid.x == 0) return 0;
  const size_t lds = sizeof(float) * (kHyperCap + kWavesPerBlock * kScratchFloatsPerWave);
  return dispatch_cfg(g.f, bwd_vec4(g, K, V, grad_out, dQ, dQ), [&](auto cfg) {
    using C = decltype(cfg);
    gt_bwd_rows_kernel<C><<<grid, kBlock, lds, s>>>(g, K, V, attn_edge, grad_out, grad_edge, dQ, chunks);
    return launch_status();
  });
}

int launch_gt_bwd_cols(const Csr &g, const int *col_ptr, const int *row_ind, const int *val_idx, const float *Q,
                       const float *attn_edge, const float *grad_edge, const float *grad_out, float *dK,
                       float *dV, const int *chunks, int nchunks, hipStream_t s) {
  const dim3 grid(chunks ? nchunks : (g.m + kWavesPerBlock - 1) / kWavesPerBlock, g.h);
  if (grid.x == 0) return 0;
  return dispatch_cfg(g.f, bwd_vec4(g, Q, grad_out, dK, dV, dV), [&](auto cfg) {
    using C = decltype(cfg);
    gt_bwd_cols_kernel<C><<<grid, kBlock, 0, s>>>(g, col_ptr, row_ind, val_idx, Q, attn_edge, grad_edge,
                                                  grad_out, dK, dV, chunks);
    return launch_status();
  });
}

}  // namespace dfgnn
