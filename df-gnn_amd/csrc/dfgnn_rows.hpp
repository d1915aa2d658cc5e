// dfgnn_rows.hpp -- wave-cooperative row routines (softmax + SpMM) shared by the GT and GAT kernels.
//
//  * sddmm_range          edge-parallel <lhs[row], rhs[col]> for a contiguous edge range, one group
//                         of lanes per edge, 4 edges in flight per group.
//  * row_softmax_spmm_lds the node-parallel phase of the 'hyper'/'softmax' variants: the row's
//                         logits already sit in LDS.
//  * gt_row_online /      the 'tiling' variant: 64-edge tiles + online softmax, no degree limit.
//    gat_row_online       Also the fallback of the hyper kernels for workgroups whose edges do
//                         not fit the LDS budget.
#pragma once
#include "dfgnn_device.hpp"

namespace dfgnn {

// For edges e = ebeg+first, ebeg+first+stride, ... < eend: d_e = <A[ra[e]], B[cb[e]]> * val[e];
// lane 0 of the group calls wr(e, d_e).  A and B are already offset to the head.
template <class C, class Writer>
__device__ __forceinline__ void sddmm_range(int ebeg, int eend, int first, int stride,
                                            const int *__restrict__ ra, const int *__restrict__ cb,
                                            const float *__restrict__ val, const float *__restrict__ A,
                                            const float *__restrict__ B, size_t hf, int f, int gl, Writer wr) {
  int e = ebeg + first;
  for (; e + 3 * stride < eend; e += 4 * stride) {
    const int e1 = e + stride, e2 = e + 2 * stride, e3 = e + 3 * stride;
    const int r0 = ra[e], r1 = ra[e1], r2 = ra[e2], r3 = ra[e3];
    const int c0 = cb[e], c1 = cb[e1], c2 = cb[e2], c3 = cb[e3];
    Frag<C> a0, a1, a2, a3, b0, b1, b2, b3;
    frag_load<C>(a0, A + (size_t)r0 * hf, f, gl);
    frag_load<C>(b0, B + (size_t)c0 * hf, f, gl);
    frag_load<C>(a1, A + (size_t)r1 * hf, f, gl);
    frag_load<C>(b1, B + (size_t)c1 * hf, f, gl);
    frag_load<C>(a2, A + (size_t)r2 * hf, f, gl);
    frag_load<C>(b2, B + (size_t)c2 * hf, f, gl);
    frag_load<C>(a3, A + (size_t)r3 * hf, f, gl);
    frag_load<C>(b3, B + (size_t)c3 * hf, f, gl);
    float d0 = frag_dot<C>(a0, b0), d1 = frag_dot<C>(a1, b1), d2 = frag_dot<C>(a2, b2), d3 = frag_dot<C>(a3, b3);
    d0 = lanes_sum<C::G>(d0);
    d1 = lanes_sum<C::G>(d1);
    d2 = lanes_sum<C::G>(d2);
    d3 = lanes_sum<C::G>(d3);
    if (gl == 0) {
      wr(e, val ? d0 * val[e] : d0);
      wr(e1, val ? d1 * val[e1] : d1);
      wr(e2, val ? d2 * val[e2] : d2);
      wr(e3, val ? d3 * val[e3] : d3);
    }
  }
  for (; e < eend; e += stride) {
    Frag<C> a0, b0;
    frag_load<C>(a0, A + (size_t)ra[e] * hf, f, gl);
    frag_load<C>(b0, B + (size_t)cb[e] * hf, f, gl);
    const float d0 = lanes_sum<C::G>(frag_dot<C>(a0, b0));
    if (gl == 0) wr(e, val ? d0 * val[e] : d0);
  }
}

// Softmax over lw[0..deg) (LDS, this wave's row) followed by out_row = sum_e P_e X[cols[e]].
// lw is overwritten with exp(s - max).  attn_out (nullable) receives the normalised P_e.
template <class C, bool WRITE_ATTN>
__device__ __forceinline__ void row_softmax_spmm_lds(float *lw, int deg, const int *__restrict__ cols,
                                                     const float *__restrict__ X, size_t hf, int f,
                                                     float *__restrict__ out_row, float *__restrict__ attn_out,
                                                     int lane) {
  const int gid = lane / C::G, gl = lane % C::G;
  float mx = -INFINITY;
  for (int e = lane; e < deg; e += kWave) mx = fmaxf(mx, lw[e]);
  mx = lanes_max<kWave>(mx);
  float sum = 0.f;
  for (int e = lane; e < deg; e += kWave) {
    const float s = lw[e];
    const float p = (s == -INFINITY) ? 0.f : fast_exp(s - mx);
    lw[e] = p;
    sum += p;
  }
  sum = lanes_sum<kWave>(sum);
  const float inv = (sum != 0.f) ? 1.f / sum : 0.f;  // empty row -> 0 (fused_gtconv_hyper.cu:143)
  if constexpr (WRITE_ATTN) {
    for (int e = lane; e < deg; e += kWave) attn_out[e] = lw[e] * inv;
  }
  wave_sync();
  Frag<C> acc;
  frag_zero<C>(acc);
  spmm_accum<C>(acc, lw, cols, deg, X, hf, f, gid, gl);
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, inv, out_row, f, gl);
}

// Normalise raw logits previously parked in attn_out[0..deg) into probabilities.
__device__ __forceinline__ void normalise_parked_logits(float *__restrict__ attn_out, int deg, float m_run,
                                                        float inv, int lane) {
  for (int e = lane; e < deg; e += kWave) {
    const float s = attn_out[e];
    attn_out[e] = (s == -INFINITY) ? 0.f : fast_exp(s - m_run) * inv;
  }
}

// GT row, online softmax over 64-edge tiles.  sw/sc: this wave's 64-float / 64-int LDS scratch.
// Kh/Vh/q are already offset to the head.  attn_out (nullable): normalised P_e of the row.
template <class C, bool WRITE_ATTN>
__device__ __forceinline__ void gt_row_online(int lb, int deg, const int *__restrict__ col_ind,
                                              const float *__restrict__ val, const Frag<C> &q,
                                              const float *__restrict__ Kh, const float *__restrict__ Vh,
                                              size_t hf, int f, float *sw, int *sc, float *__restrict__ out_row,
                                              float *__restrict__ attn_out, int lane) {
  const int gid = lane / C::G, gl = lane % C::G;
  Frag<C> acc;
  frag_zero<C>(acc);
  float m_run = -INFINITY, l_run = 0.f;
  for (int t0 = 0; t0 < deg; t0 += kWave) {
    const int nt = min(kWave, deg - t0);
    sc[lane] = (lane < nt) ? col_ind[lb + t0 + lane] : 0;
    wave_sync();
    for (int e = gid; e < nt; e += C::EPW) {
      Frag<C> k;
      frag_load<C>(k, Kh + (size_t)sc[e] * hf, f, gl);
      const float d = lanes_sum<C::G>(frag_dot<C>(q, k));
      if (gl == 0) sw[e] = val ? d * val[lb + t0 + e] : d;
    }
    wave_sync();
    const float s = (lane < nt) ? sw[lane] : -INFINITY;
    if constexpr (WRITE_ATTN) {
      if (lane < nt) attn_out[t0 + lane] = s;  // parked raw logit, normalised after the last tile
    }
    online_step<C>(s, lane, sw, acc, m_run, l_run);
    wave_sync();
    spmm_accum<C>(acc, sw, sc, nt, Vh, hf, f, gid, gl);
    wave_sync();
  }
  const float inv = (l_run != 0.f) ? 1.f / l_run : 0.f;
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, inv, out_row, f, gl);
  if constexpr (WRITE_ATTN) normalise_parked_logits(attn_out, deg, m_run, inv, lane);
}

// GAT row, online softmax; logits are one lane per edge: LeakyReLU(ar + attn_col[col*h]).
template <class C>
__device__ __forceinline__ void gat_row_online(int lb, int deg, const int *__restrict__ col_ind, float ar,
                                               const float *__restrict__ attn_col_h, int h, float slope,
                                               const float *__restrict__ Xh, size_t hf, int f, float *sw, int *sc,
                                               float *__restrict__ out_row, int lane) {
  const int gid = lane / C::G, gl = lane % C::G;
  Frag<C> acc;
  frag_zero<C>(acc);
  float m_run = -INFINITY, l_run = 0.f;
  for (int t0 = 0; t0 < deg; t0 += kWave) {
    const int nt = min(kWave, deg - t0);
    float s = -INFINITY;
    int c = 0;
    if (lane < nt) {
      c = col_ind[lb + t0 + lane];
      s = leaky_relu(ar + attn_col_h[(size_t)c * h], slope);
    }
    sc[lane] = c;
    online_step<C>(s, lane, sw, acc, m_run, l_run);
    wave_sync();
    spmm_accum<C>(acc, sw, sc, nt, Xh, hf, f, gid, gl);
    wave_sync();
  }
  const float inv = (l_run != 0.f) ? 1.f / l_run : 0.f;
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, inv, out_row, f, gl);
}

}  // namespace dfgnn
