// softmax_spmm.hip -- second kernel of the two-kernel 'softmax' / 'softmax_gm' variants (GT and GAT):
// node-parallel softmax + SpMM over logits that the edge-parallel first kernel left in global memory.
//
//   USE_LDS = true  ('softmax')    the wave copies its row's logits into LDS once, then works from LDS.
//                                  replaces softMax_SPMM (DFGNN/src/spmm/spmm.cuh:7-83), whose 128-float
//                                  row buffer overflows for degree > 128: rows longer
//                                  than kRowCap fall through to the global-memory passes instead.
//   USE_LDS = false ('softmax_gm') three passes over the logits in global memory (max, sum, SpMM).
//                                  replaces softMax_SPMM_global_memory (spmm.cuh:85-150,
//                                  fused_gtconv_softmax_gm.cu:9-79).
// Logits are [h, nnz] head-major (the reference's layouts only work for h == 1).
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

constexpr int kRowCap = 1024;  // floats of LDS per wave for the 'softmax' variant

template <class C, bool USE_LDS>
__global__ __launch_bounds__(kBlock) void softmax_spmm_kernel(Csr g, const float *__restrict__ logits,
                                                              const float *__restrict__ X,
                                                              float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[USE_LDS ? kWavesPerBlock * kRowCap : 4];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f;
  const int f = g.f;
  const float *Xh = X + (size_t)head * f;
  const float *lg_h = logits + (size_t)head * g.nnz;
  const int gid = lane / C::G, gl = lane % C::G;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < g.m; r += gridDim.x * kWavesPerBlock) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    float *out_row = out + (size_t)r * hf + (size_t)head * f;
    const float *lg = lg_h + lb;
    if (USE_LDS && deg <= kRowCap) {
      float *lw = lds + wave * kRowCap;
      for (int e = lane; e < deg; e += kWave) lw[e] = lg[e];
      wave_sync();
      row_softmax_spmm_lds<C, false>(lw, deg, g.col_ind + lb, Xh, hf, f, out_row, nullptr, lane);
      wave_sync();
    } else {
      float mx = -INFINITY;
      for (int e = lane; e < deg; e += kWave) mx = fmaxf(mx, lg[e]);
      mx = lanes_max<kWave>(mx);
      float sum = 0.f;
      for (int e = lane; e < deg; e += kWave) {
        const float s = lg[e];
        sum += (s == -INFINITY) ? 0.f : fast_exp(s - mx);
      }
      sum = lanes_sum<kWave>(sum);
      const float inv = (sum != 0.f) ? 1.f / sum : 0.f;
      Frag<C> acc;
      frag_zero<C>(acc);
      spmm_accum<C>(acc, lg, g.col_ind + lb, deg, Xh, hf, f, gid, gl,
                    [mx](float s) { return (s == -INFINITY) ? 0.f : fast_exp(s - mx); });
      frag_reduce_groups<C>(acc);
      if (gid == 0) frag_store_scaled<C>(acc, inv, out_row, f, gl);
    }
  }
}

int launch_softmax_spmm(const Csr &g, const float *logits, const float *X, float *out, bool use_lds,
                        hipStream_t s) {
  const int blocks = (g.m + kWavesPerBlock - 1) / kWavesPerBlock;
  const dim3 grid(blocks, g.h);
  const bool v4 = (g.f % 4 == 0) && aligned16(X) && aligned16(out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    if (use_lds)
      softmax_spmm_kernel<C, true><<<grid, kBlock, 0, s>>>(g, logits, X, out);
    else
      softmax_spmm_kernel<C, false><<<grid, kBlock, 0, s>>>(g, logits, X, out);
    return launch_status();
  });
}

}  // namespace dfgnn
