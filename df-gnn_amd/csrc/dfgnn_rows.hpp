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

// GT backward, CSR pass for one row by one wave, any degree: dP_e = <dO_r, V_c> in 64-edge tiles (parked in dS_h
// between the two sweeps), dS_e = P_e (dP_e - sum_row P dP) -> dS_h, dQ_r = sum_e dS_e val_e K_c.  Pointers are offset
// to the head; P_h / dS_h are the head's [nnz] arrays.
template <class C>
__device__ __forceinline__ void gt_bwd_row_online(int r, int lb, int deg, const int *__restrict__ col_ind,
                                                  const float *__restrict__ val, const float *__restrict__ Kh,
                                                  const float *__restrict__ Vh, const float *__restrict__ dOh,
                                                  const float *__restrict__ P_h, float *__restrict__ dS_h, size_t hf,
                                                  int f, float *sw, int *sc, float *__restrict__ dQh, int lane) {
  const int gid = lane / C::G, gl = lane % C::G;
  Frag<C> go;
  frag_load<C>(go, dOh + (size_t)r * hf, f, gl);
  float t = 0.f;
  for (int t0 = 0; t0 < deg; t0 += kWave) {
    const int nt = min(kWave, deg - t0);
    sc[lane] = (lane < nt) ? col_ind[lb + t0 + lane] : 0;
    wave_sync();
    for (int e = gid; e < nt; e += C::EPW) {
      Frag<C> v;
      frag_load<C>(v, Vh + (size_t)sc[e] * hf, f, gl);
      const float d = lanes_sum<C::G>(frag_dot<C>(go, v));
      if (gl == 0) sw[e] = d;
    }
    wave_sync();
    if (lane < nt) {
      const float dp = sw[lane];
      dS_h[lb + t0 + lane] = dp;
      t = fmaf(dp, P_h[lb + t0 + lane], t);
    }
    wave_sync();
  }
  t = lanes_sum<kWave>(t);
  Frag<C> acc;
  frag_zero<C>(acc);
  for (int t0 = 0; t0 < deg; t0 += kWave) {
    const int nt = min(kWave, deg - t0);
    float w = 0.f;
    int c = 0;
    if (lane < nt) {
      const int e = lb + t0 + lane;
      const float ds = P_h[e] * (dS_h[e] - t);  // same lane wrote dS_h[e] above
      dS_h[e] = ds;
      w = val ? ds * val[e] : ds;
      c = col_ind[e];
    }
    sw[lane] = w;
    sc[lane] = c;
    wave_sync();
    spmm_accum<C>(acc, sw, sc, nt, Kh, hf, f, gid, gl);
    wave_sync();
  }
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, 1.f, dQh + (size_t)r * hf, f, gl);
}

// GT backward, CSC pass for one column by one wave: dV_j = sum P_e dO_i, dK_j = sum dS_e val_e Q_i over the column's
// CSC entries (lb .. lb + n), two entries in flight per lane group.
template <class C>
__device__ __forceinline__ void gt_bwd_col_wave(int j, int lb, int n, const int *__restrict__ row_ind,
                                                const int *__restrict__ val_idx, const float *__restrict__ val,
                                                const float *__restrict__ Qh, const float *__restrict__ dOh,
                                                const float *__restrict__ P_h, const float *__restrict__ dS_h, size_t hf,
                                                int f, float *__restrict__ dKh, float *__restrict__ dVh, int lane) {
  const int gid = lane / C::G, gl = lane % C::G;
  Frag<C> aK, aV;
  frag_zero<C>(aK);
  frag_zero<C>(aV);
  int t = gid;
  for (; t + C::EPW < n; t += 2 * C::EPW) {
    const int ea = val_idx[lb + t], eb = val_idx[lb + t + C::EPW];
    const int ia = row_ind[lb + t], ib = row_ind[lb + t + C::EPW];
    const float pa = P_h[ea], pb = P_h[eb];
    const float sa = val ? dS_h[ea] * val[ea] : dS_h[ea];
    const float sb = val ? dS_h[eb] * val[eb] : dS_h[eb];
    Frag<C> ga, gb, qa, qb;
    frag_load<C>(ga, dOh + (size_t)ia * hf, f, gl);
    frag_load<C>(qa, Qh + (size_t)ia * hf, f, gl);
    frag_load<C>(gb, dOh + (size_t)ib * hf, f, gl);
    frag_load<C>(qb, Qh + (size_t)ib * hf, f, gl);
    frag_fma<C>(aV, pa, ga);
    frag_fma<C>(aK, sa, qa);
    frag_fma<C>(aV, pb, gb);
    frag_fma<C>(aK, sb, qb);
  }
  for (; t < n; t += C::EPW) {
    const int ea = val_idx[lb + t], ia = row_ind[lb + t];
    const float pa = P_h[ea];
    const float sa = val ? dS_h[ea] * val[ea] : dS_h[ea];
    Frag<C> ga, qa;
    frag_load<C>(ga, dOh + (size_t)ia * hf, f, gl);
    frag_load<C>(qa, Qh + (size_t)ia * hf, f, gl);
    frag_fma<C>(aV, pa, ga);
    frag_fma<C>(aK, sa, qa);
  }
  frag_reduce_groups<C>(aK);
  frag_reduce_groups<C>(aV);
  if (gid == 0) {
    frag_store_scaled<C>(aK, 1.f, dKh + (size_t)j * hf, f, gl);
    frag_store_scaled<C>(aV, 1.f, dVh + (size_t)j * hf, f, gl);
  }
}

}  // namespace dfgnn
