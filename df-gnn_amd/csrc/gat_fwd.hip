// gat_fwd.hip -- GAT forward kernels for gfx950.
//
//   gat_hyper_fwd_kernel   'hyper' : one kernel, CSR+COO.  Phase 1 thread-per-edge logits
//                                    LeakyReLU(attn_row[src] + attn_col[dst]) into LDS, phase 2 a wave per
//                                    row (softmax + SpMM).  replaces fused_gat_hyper_inference{,_vec4}
//                                    (DFGNN/src/fused_gatconv/fused_gatconv_hyper.cu:5-224)
//   gat_tiling_fwd_kernel  'tiling': one kernel, CSR only, a wave per row, 64-edge tiles, online softmax.
//                                    replaces fused_gat_tiling (fused_gatconv_tiling.cu:9-76), including its
//                                    f > 128 / f % 32 != 0 failure modes.
//   gat_sddmm_kernel       first kernel of 'softmax'/'softmax_gm': edge-parallel logits to global memory,
//                                    all heads (the reference's gat_sddmmCooKernel, sddmm.cuh:7-32, computes
//                                    head 0 only).
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

template <class C>
__global__ __launch_bounds__(kBlock) void gat_hyper_fwd_kernel(Csr g, const float *__restrict__ attn_row,
                                                               const float *__restrict__ attn_col, float slope,
                                                               const float *__restrict__ X,
                                                               float *__restrict__ out,
                                                               const int *__restrict__ chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *lw = lds;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + kHyperCap + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);

  const int head = blockIdx.y;
  const int h = g.h, f = g.f;
  const int r0 = chunks ? chunks[2 * blockIdx.x] : blockIdx.x * kHyperRows;
  const int r1 = chunks ? chunks[2 * blockIdx.x + 1] : min(g.m, r0 + kHyperRows);
  const size_t hf = (size_t)h * f;
  const float *Xh = X + (size_t)head * f;
  const float *arow_h = attn_row + head, *acol_h = attn_col + head;
  float *outh = out + (size_t)head * f;
  const int e0 = g.row_ptr[r0], e1 = g.row_ptr[r1];
  const int ne = e1 - e0;

  if (ne <= kHyperCap) {
    for (int e = e0 + (int)threadIdx.x; e < e1; e += kBlock) {
      const int src = g.rows[e], dst = g.col_ind[e];
      lw[e - e0] = leaky_relu(arow_h[(size_t)src * h] + acol_h[(size_t)dst * h], slope);
    }
    __syncthreads();
    for (int r = r0 + wave; r < r1; r += kWavesPerBlock) {
      const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
      row_softmax_spmm_lds<C, false>(lw + (lb - e0), deg, g.col_ind + lb, Xh, hf, f, outh + (size_t)r * hf,
                                     nullptr, lane);
    }
  } else {
    for (int r = r0 + wave; r < r1; r += kWavesPerBlock) {
      const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
      gat_row_online<C>(lb, deg, g.col_ind, arow_h[(size_t)r * h], acol_h, h, slope, Xh, hf, f, sw, sc,
                        outh + (size_t)r * hf, lane);
    }
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gat_tiling_fwd_kernel(Csr g, const float *__restrict__ attn_row,
                                                                const float *__restrict__ attn_col, float slope,
                                                                const float *__restrict__ X,
                                                                float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  const int head = blockIdx.y;
  const int h = g.h, f = g.f;
  const size_t hf = (size_t)h * f;
  const float *Xh = X + (size_t)head * f;
  const float *arow_h = attn_row + head, *acol_h = attn_col + head;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < g.m; r += gridDim.x * kWavesPerBlock) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    gat_row_online<C>(lb, deg, g.col_ind, arow_h[(size_t)r * h], acol_h, h, slope, Xh, hf, f, sw, sc,
                      out + (size_t)r * hf + (size_t)head * f, lane);
  }
}

__global__ __launch_bounds__(kBlock) void gat_sddmm_kernel(Csr g, const float *__restrict__ attn_row,
                                                           const float *__restrict__ attn_col, float slope,
                                                           float *__restrict__ logits) {
  const int head = blockIdx.y, h = g.h;
  float *lg = logits + (size_t)head * g.nnz;
  for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < g.nnz; e += (long)gridDim.x * kBlock) {
    const int src = g.rows[e], dst = g.col_ind[e];
    lg[e] = leaky_relu(attn_row[(size_t)src * h + head] + attn_col[(size_t)dst * h + head], slope);
  }
}

// attn_row[i, hd] = <a_l[hd, :], X[i, hd, :]>, attn_col[i, hd] = <a_r[hd, :], X[i, hd, :]>: X is read once for both.
// A lane group per (node, head).  replaces fused_gat_dot_attn_weight (fused_gatconv_hyper_v2.cu:212-249), the first
// kernel of the 'hyper_v2' variant.
template <class C>
__global__ __launch_bounds__(kBlock) void gat_attn_scores_kernel(int m, int h, int f, const float *__restrict__ a_l,
                                                                 const float *__restrict__ a_r,
                                                                 const float *__restrict__ X,
                                                                 float *__restrict__ attn_row,
                                                                 float *__restrict__ attn_col) {
  const int gl = threadIdx.x % C::G;
  const long stride = (long)gridDim.x * (kBlock / C::G);
  Frag<C> al, ar;
  const int head = blockIdx.y;
  frag_load<C>(al, a_l + (size_t)head * f, f, gl);
  frag_load<C>(ar, a_r + (size_t)head * f, f, gl);
  for (long i = (long)blockIdx.x * (kBlock / C::G) + threadIdx.x / C::G; i < m; i += stride) {
    Frag<C> x;
    frag_load<C>(x, X + ((size_t)i * h + head) * f, f, gl);
    const float r = lanes_sum<C::G>(frag_dot<C>(al, x)), c = lanes_sum<C::G>(frag_dot<C>(ar, x));
    if (gl == 0) {
      attn_row[(size_t)i * h + head] = r;
      attn_col[(size_t)i * h + head] = c;
    }
  }
}

int launch_gat_attn_scores(int m, int h, int f, const float *a_l, const float *a_r, const float *X, float *attn_row,
                           float *attn_col, hipStream_t s) {
  const bool v4 = (f % 4 == 0) && aligned16(X) && aligned16(a_l) && aligned16(a_r);
  return dispatch_cfg(f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    const long per = kBlock / C::G;
    long blocks = ((long)m + per - 1) / per;
    if (blocks > 65536) blocks = 65536;
    gat_attn_scores_kernel<C><<<dim3((unsigned)blocks, h), kBlock, 0, s>>>(m, h, f, a_l, a_r, X, attn_row, attn_col);
    return launch_status();
  });
}

int launch_gat_hyper_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope,
                         const float *X, float *out, const int *chunks, int nchunks, hipStream_t s) {
  const dim3 grid(chunks ? nchunks : (g.m + kHyperRows - 1) / kHyperRows, g.h);
  if (grid.x == 0) return 0;
  const size_t lds = sizeof(float) * (kHyperCap + kWavesPerBlock * kScratchFloatsPerWave);
  const bool v4 = (g.f % 4 == 0) && aligned16(X) && aligned16(out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    gat_hyper_fwd_kernel<C><<<grid, kBlock, lds, s>>>(g, attn_row, attn_col, slope, X, out, chunks);
    return launch_status();
  });
}

int launch_gat_tiling_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope,
                          const float *X, float *out, hipStream_t s) {
  const dim3 grid((g.m + kWavesPerBlock - 1) / kWavesPerBlock, g.h);
  const bool v4 = (g.f % 4 == 0) && aligned16(X) && aligned16(out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    gat_tiling_fwd_kernel<C><<<grid, kBlock, 0, s>>>(g, attn_row, attn_col, slope, X, out);
    return launch_status();
  });
}

int launch_gat_sddmm(const Csr &g, const float *attn_row, const float *attn_col, float slope, float *logits,
                     hipStream_t s) {
  if (g.nnz == 0) return 0;
  long want = ((long)g.nnz + kBlock * 4 - 1) / (kBlock * 4);
  const int blocks = (int)(want < 1 ? 1 : (want > 32768 ? 32768 : want));
  gat_sddmm_kernel<<<dim3(blocks, g.h), kBlock, 0, s>>>(g, attn_row, attn_col, slope, logits);
  return launch_status();
}

}  // namespace dfgnn
