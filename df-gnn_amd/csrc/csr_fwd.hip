// csr_fwd.hip -- the node-parallel CSR baselines of the reference's comparison sweeps, as kernels of their own.
//
//   gt_csr_fwd_kernel<USE_LDS>   'csr' / 'csr_gm' (GT): one WAVE per row (the reference: one block per row).  Sweep 1
//                                computes every logit of the row -- a group of lanes per edge, Q_i in registers -- and
//                                parks it in this wave's LDS row buffer ('csr'; rows longer than the buffer use the
//                                global scratch, the reference overflows its 128-float buffer) or in global memory
//                                ('csr_gm'); then max, sum of exp and the weighted sum run over the parked logits.
//                                replaces fused_gt_csr / fused_gt_csr_global_memory
//                                (DFGNN/src/fused_gtconv/fused_gtconv_csr.cu:10-113, :115-219)
//   gat_recompute_fwd_kernel     'hyper_recompute' (GAT): one wave per row, NO logit storage: the rank-one logits
//                                LeakyReLU(attn_row[i] + attn_col[j]) are recomputed in each of the three sweeps (max,
//                                sum, weighted sum).  replaces fused_gat_hyper_recompute_inference_vec4
//                                (DFGNN/src/fused_gatconv/fused_gatconv_hyper_recompute.cu:118-216), any f (the reference
//                                exit(0)s unless f % 128 == 0).
// Unlike the 'tiling' kernels (online softmax, one sweep) these are the three-sweep algorithms the paper compares against;
// `--format all` sweeps therefore time different kernels under the different format names.
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

constexpr int kCsrRowCap = 2048;  // floats of LDS per wave for the row's logits ('csr')

template <class C, bool USE_LDS>
__global__ __launch_bounds__(kBlock) void gt_csr_fwd_kernel(Csr g, const float *__restrict__ Q,
                                                            const float *__restrict__ K, const float *__restrict__ V,
                                                            float *__restrict__ logits, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[USE_LDS ? kWavesPerBlock * kCsrRowCap : 4];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int gid = lane / C::G, gl = lane % C::G;
  const int head = blockIdx.y, f = g.f;
  const size_t hf = (size_t)g.h * f;
  const float *Qh = Q + (size_t)head * f, *Kh = K + (size_t)head * f, *Vh = V + (size_t)head * f;
  float *lg_h = logits + (size_t)head * g.nnz;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < g.m; r += gridDim.x * kWavesPerBlock) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    const int *cols = g.col_ind + lb;
    float *out_row = out + (size_t)r * hf + (size_t)head * f;
    const bool in_lds = USE_LDS && deg <= kCsrRowCap;
    float *lw = in_lds ? lds + wave * kCsrRowCap : lg_h + lb;
    // sweep 1: s_e = val_e <Q_i, K_j>, one lane group per edge, four gathers in flight per group
    Frag<C> q;
    frag_load<C>(q, Qh + (size_t)r * hf, f, gl);
    int e = gid;
    for (; e + 3 * C::EPW < deg; e += 4 * C::EPW) {
      Frag<C> k0, k1, k2, k3;
      frag_load<C>(k0, Kh + (size_t)cols[e] * hf, f, gl);
      frag_load<C>(k1, Kh + (size_t)cols[e + C::EPW] * hf, f, gl);
      frag_load<C>(k2, Kh + (size_t)cols[e + 2 * C::EPW] * hf, f, gl);
      frag_load<C>(k3, Kh + (size_t)cols[e + 3 * C::EPW] * hf, f, gl);
      const float d0 = lanes_sum<C::G>(frag_dot<C>(q, k0)), d1 = lanes_sum<C::G>(frag_dot<C>(q, k1)),
                  d2 = lanes_sum<C::G>(frag_dot<C>(q, k2)), d3 = lanes_sum<C::G>(frag_dot<C>(q, k3));
      if (gl == 0) {
        lw[e] = g.val ? d0 * g.val[lb + e] : d0;
        lw[e + C::EPW] = g.val ? d1 * g.val[lb + e + C::EPW] : d1;
        lw[e + 2 * C::EPW] = g.val ? d2 * g.val[lb + e + 2 * C::EPW] : d2;
        lw[e + 3 * C::EPW] = g.val ? d3 * g.val[lb + e + 3 * C::EPW] : d3;
      }
    }
    for (; e < deg; e += C::EPW) {
      Frag<C> k0;
      frag_load<C>(k0, Kh + (size_t)cols[e] * hf, f, gl);
      const float d0 = lanes_sum<C::G>(frag_dot<C>(q, k0));
      if (gl == 0) lw[e] = g.val ? d0 * g.val[lb + e] : d0;
    }
    if (in_lds) {
      wave_sync();
      row_softmax_spmm_lds<C, false>(lw, deg, cols, Vh, hf, f, out_row, nullptr, lane);  // sweeps 2-4 from LDS
      wave_sync();
    } else {
      // the logits went to global memory: this wave re-reads what other lanes of it wrote
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      wave_sync();
      float mx = -INFINITY;
      for (int k = lane; k < deg; k += kWave) mx = fmaxf(mx, lw[k]);
      mx = lanes_max<kWave>(mx);
      float sum = 0.f;
      for (int k = lane; k < deg; k += kWave) {
        const float s = lw[k];
        sum += (s == -INFINITY) ? 0.f : fast_exp(s - mx);
      }
      sum = lanes_sum<kWave>(sum);
      const float inv = (sum != 0.f) ? 1.f / sum : 0.f;
      Frag<C> acc;
      frag_zero<C>(acc);
      spmm_accum<C>(acc, lw, cols, deg, Vh, hf, f, gid, gl,
                    [mx](float s) { return (s == -INFINITY) ? 0.f : fast_exp(s - mx); });
      frag_reduce_groups<C>(acc);
      if (gid == 0) frag_store_scaled<C>(acc, inv, out_row, f, gl);
    }
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gat_recompute_fwd_kernel(Csr g, const float *__restrict__ attn_row,
                                                                   const float *__restrict__ attn_col, float slope,
                                                                   const float *__restrict__ X,
                                                                   float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  const int gid = lane / C::G, gl = lane % C::G;
  const int head = blockIdx.y, f = g.f, h = g.h;
  const size_t hf = (size_t)h * f;
  const float *Xh = X + (size_t)head * f, *ac = attn_col + head;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < g.m; r += gridDim.x * kWavesPerBlock) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    const int *cols = g.col_ind + lb;
    const float ar = attn_row[(size_t)r * h + head];
    auto logit = [&](int k) { return leaky_relu(ar + ac[(size_t)cols[k] * h], slope); };
    float mx = -INFINITY;  // sweep 1: max
    for (int k = lane; k < deg; k += kWave) mx = fmaxf(mx, logit(k));
    mx = lanes_max<kWave>(mx);
    float sum = 0.f;  // sweep 2: sum of exp (logits recomputed)
    for (int k = lane; k < deg; k += kWave) sum += fast_exp(logit(k) - mx);
    sum = lanes_sum<kWave>(sum);
    const float inv = (sum != 0.f) ? 1.f / sum : 0.f;
    Frag<C> acc;  // sweep 3: weighted sum (logits recomputed a third time, 64 edges at a time)
    frag_zero<C>(acc);
    for (int t0 = 0; t0 < deg; t0 += kWave) {
      const int nt = min(kWave, deg - t0);
      sc[lane] = (lane < nt) ? cols[t0 + lane] : 0;
      sw[lane] = (lane < nt) ? fast_exp(logit(t0 + lane) - mx) : 0.f;
      wave_sync();
      spmm_accum<C>(acc, sw, sc, nt, Xh, hf, f, gid, gl);
      wave_sync();
    }
    frag_reduce_groups<C>(acc);
    if (gid == 0) frag_store_scaled<C>(acc, inv, out + (size_t)r * hf + (size_t)head * f, f, gl);
  }
}

int launch_gt_csr_fwd(const Csr &g, const float *Q, const float *K, const float *V, float *logits, float *out,
                      bool use_lds, hipStream_t s) {
  const dim3 grid((g.m + kWavesPerBlock - 1) / kWavesPerBlock, g.h);
  const bool v4 = (g.f % 4 == 0) && aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    if (use_lds) gt_csr_fwd_kernel<C, true><<<grid, kBlock, 0, s>>>(g, Q, K, V, logits, out);
    else gt_csr_fwd_kernel<C, false><<<grid, kBlock, 0, s>>>(g, Q, K, V, logits, out);
    return launch_status();
  });
}

int launch_gat_recompute_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                             float *out, hipStream_t s) {
  const dim3 grid((g.m + kWavesPerBlock - 1) / kWavesPerBlock, g.h);
  const bool v4 = (g.f % 4 == 0) && aligned16(X) && aligned16(out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    gat_recompute_fwd_kernel<C><<<grid, kBlock, 0, s>>>(g, attn_row, attn_col, slope, X, out);
    return launch_status();
  });
}

}  // namespace dfgnn
