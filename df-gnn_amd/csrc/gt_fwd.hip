// gt_fwd.hip -- graph-transformer forward kernels for gfx950.
//
//   gt_hyper_fwd_kernel   'hyper'  : ONE kernel over the CSR+COO format.  Phase 1 is edge-parallel
//                                    (a group of lanes per edge, logits into LDS), phase 2 is
//                                    node-parallel (a wave per row: softmax + SpMM).
//                                    replaces fused_gt_hyper / fused_gt_hyper_inference{,_vec4,_small_f}
//                                    (DFGNN/src/fused_gtconv/fused_gtconv_hyper.cu:31-532)
//   gt_tiling_fwd_kernel  'tiling' : ONE kernel, CSR only, a wave per row, 64-edge tiles with
//                                    online softmax.  replaces fused_gt_tiling
//                                    (DFGNN/src/fused_gtconv/fused_gtconv_tiling.cu:9-90)
//   gt_sddmm_kernel       first kernel of the two-kernel 'softmax' variant.  replaces
//                                    sddmmCooKernel (DFGNN/src/sddmm/sddmm.cuh:34-71)
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

template <class C, bool WRITE_ATTN>
__global__ __launch_bounds__(kBlock) void gt_hyper_fwd_kernel(Csr g, const float *__restrict__ Q,
                                                              const float *__restrict__ K,
                                                              const float *__restrict__ V,
                                                              float *__restrict__ attn_edge,
                                                              float *__restrict__ out,
                                                              const int *__restrict__ chunks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *lw = lds;                                    // [kHyperCap] logits of this workgroup's edges
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + kHyperCap + wave * kScratchFloatsPerWave;  // per-wave scratch (online fallback)
  int *sc = reinterpret_cast<int *>(sw + kWave);

  const int head = blockIdx.y;
  // rows of this workgroup: a fixed 16-row slice, or (with a block plan) one spill chunk
  const int r0 = chunks ? chunks[2 * blockIdx.x] : blockIdx.x * kHyperRows;
  const int r1 = chunks ? chunks[2 * blockIdx.x + 1] : min(g.m, r0 + kHyperRows);
  const size_t hf = (size_t)g.h * g.f;
  const int f = g.f;
  const float *Qh = Q + (size_t)head * f, *Kh = K + (size_t)head * f, *Vh = V + (size_t)head * f;
  float *outh = out + (size_t)head * f;
  float *attn_h = WRITE_ATTN ? attn_edge + (size_t)head * g.nnz : nullptr;

  const int e0 = g.row_ptr[r0], e1 = g.row_ptr[r1];
  const int ne = e1 - e0;
  const int gl = lane % C::G;

  if (ne <= kHyperCap) {
    // phase 1: edge-parallel SDDMM over this workgroup's contiguous edge range
    constexpr int NG = kBlock / C::G;
    const int ggid = threadIdx.x / C::G;
    sddmm_range<C>(e0, e1, ggid, NG, g.rows, g.col_ind, g.val, Qh, Kh, hf, f, gl,
                   [&](int e, float s) { lw[e - e0] = s; });
    __syncthreads();
    // phase 2: node-parallel softmax + SpMM, one wave per row
    for (int r = r0 + wave; r < r1; r += kWavesPerBlock) {
      const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
      row_softmax_spmm_lds<C, WRITE_ATTN>(lw + (lb - e0), deg, g.col_ind + lb, Vh, hf, f, outh + (size_t)r * hf,
                                          WRITE_ATTN ? attn_h + lb : nullptr, lane);
    }
  } else {
    // rows too heavy for the LDS budget: online softmax per row, no degree limit
    for (int r = r0 + wave; r < r1; r += kWavesPerBlock) {
      const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
      Frag<C> q;
      frag_load<C>(q, Qh + (size_t)r * hf, f, gl);
      gt_row_online<C, WRITE_ATTN>(lb, deg, g.col_ind, g.val, q, Kh, Vh, hf, f, sw, sc, outh + (size_t)r * hf,
                                   WRITE_ATTN ? attn_h + lb : nullptr, lane);
    }
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gt_tiling_fwd_kernel(Csr g, const float *__restrict__ Q,
                                                               const float *__restrict__ K,
                                                               const float *__restrict__ V,
                                                               float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f;
  const int f = g.f;
  const float *Qh = Q + (size_t)head * f, *Kh = K + (size_t)head * f, *Vh = V + (size_t)head * f;
  const int gl = lane % C::G;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < g.m; r += gridDim.x * kWavesPerBlock) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    Frag<C> q;
    frag_load<C>(q, Qh + (size_t)r * hf, f, gl);
    gt_row_online<C, false>(lb, deg, g.col_ind, g.val, q, Kh, Vh, hf, f, sw, sc,
                            out + (size_t)r * hf + (size_t)head * f, nullptr, lane);
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gt_sddmm_kernel(Csr g, const float *__restrict__ Q,
                                                          const float *__restrict__ K,
                                                          float *__restrict__ logits) {
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f;
  constexpr int NG = kBlock / C::G;
  const int ggid = blockIdx.x * NG + threadIdx.x / C::G;
  const int gl = threadIdx.x % C::G;
  float *lg = logits + (size_t)head * g.nnz;
  sddmm_range<C>(0, g.nnz, ggid, gridDim.x * NG, g.rows, g.col_ind, g.val, Q + (size_t)head * g.f,
                 K + (size_t)head * g.f, hf, g.f, gl, [&](int e, float s) { lg[e] = s; });
}

static inline bool vec4_ok(const Csr &g, const float *a, const float *b, const float *c, const float *d) {
  return (g.f % 4 == 0) && aligned16(a) && aligned16(b) && aligned16(c) && aligned16(d);
}

int launch_gt_hyper_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *attn_edge,
                        float *out, const int *chunks, int nchunks, hipStream_t s) {
  const dim3 grid(chunks ? nchunks : (g.m + kHyperRows - 1) / kHyperRows, g.h);
  if (grid.x == 0) return 0;
  const size_t lds = sizeof(float) * (kHyperCap + kWavesPerBlock * kScratchFloatsPerWave);
  return dispatch_cfg(g.f, vec4_ok(g, Q, K, V, out), [&](auto cfg) {
    using C = decltype(cfg);
    if (attn_edge)
      gt_hyper_fwd_kernel<C, true><<<grid, kBlock, lds, s>>>(g, Q, K, V, attn_edge, out, chunks);
    else
      gt_hyper_fwd_kernel<C, false><<<grid, kBlock, lds, s>>>(g, Q, K, V, nullptr, out, chunks);
    return launch_status();
  });
}

int launch_gt_tiling_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *out,
                         hipStream_t s) {
  const int blocks = (g.m + kWavesPerBlock - 1) / kWavesPerBlock;
  const dim3 grid(blocks, g.h);
  return dispatch_cfg(g.f, vec4_ok(g, Q, K, V, out), [&](auto cfg) {
    using C = decltype(cfg);
    gt_tiling_fwd_kernel<C><<<grid, kBlock, 0, s>>>(g, Q, K, V, out);
    return launch_status();
  });
}

int launch_gt_sddmm(const Csr &g, const float *Q, const float *K, float *logits, hipStream_t s) {
  if (g.nnz == 0) return 0;
  return dispatch_cfg(g.f, vec4_ok(g, Q, K, Q, K), [&](auto cfg) {
    using C = decltype(cfg);
    constexpr int NG = kBlock / C::G;
    long want = ((long)g.nnz + (long)NG * 4 - 1) / ((long)NG * 4);
    const int blocks = (int)(want < 1 ? 1 : (want > 32768 ? 32768 : want));
    gt_sddmm_kernel<C><<<dim3(blocks, g.h), kBlock, 0, s>>>(g, Q, K, logits);
    return launch_status();
  });
}

}  // namespace dfgnn
