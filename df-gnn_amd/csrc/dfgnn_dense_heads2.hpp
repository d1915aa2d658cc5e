// dfgnn_dense_heads2.hpp -- matrix-core GT backward for MULTI-HEAD configurations WITHOUT attn_edge: the backward of
// the statistics-saving training pair (gt_dense_stats.hip), every head of a dense range of <= 160 nodes in one
// workgroup, and no P / dS tile.
//
// The attn_edge-based backward (dfgnn_dense_heads.hpp: dense_bwd_heads_body) moves 8 h nnz bytes of attention values
// through HBM, scatters them into an n x n tile per head, reads them back, and passes P and dS through that tile to the
// column products -- four barriers, a scatter and ~70 KB of LDS writes per head.  None of that is needed when P can be
// recomputed: with the forward's row statistics (logit maximum m_i, sum of exponentials l_i) P_ij = exp(S_ij - m_i) / l_i
// is an ELEMENTWISE function of S = Q K^T, which costs 3 MFMAs per 16 x 16 tile for heads of <= 32 features.  So every
// product is computed in the orientation in which its accumulators ARE the next product's operand (dfgnn_dense.hpp: a
// D^T tile is the B operand of a product that contracts over its row index), once per orientation:
//
//   row pass (wave = 16 rows i, images K and V of the group resident, the strip's Q / dO rows as register operands):
//       S^T = K Q^T ; dP^T = V dO^T ; P ; t_i = sum_j P dP ; dS = P (dP - t)      -> dQ^T = K^T dS^T   (t_i -> LDS)
//   col pass (wave = 16 columns j, images Q and dO resident, the strip's K / V rows as register operands):
//       S = Q K^T ; dP = dO V^T ; P, dS from the statistics and t_i               -> dV^T = dO^T P ; dK^T = Q^T dS
//
// Heads are taken in GROUPS of 64 feature columns (4 / 2 / 1 heads; dfgnn_dense_heads.hpp): two 64-wide hi / lo images
// per pass (256-byte row segments from memory), four barriers per group, none per head; the next pass's images
// travel in registers meanwhile.  The register operands of a pass come straight from memory (32-byte pieces, as the
// forward fetches Q); the other pass stages the same rows as an image a few microseconds later (an L2 hit).
// A wave's work in a pass is a list of UNITS (strip, head): the heads of its own strip, plus -- ranges of 129..160 nodes
// have ten strips for eight waves -- one (strip, head) unit of the two extra strips, so that the extra strips cost every
// wave a head instead of costing two waves a whole second strip.  The units run in a rolled loop (one copy of the code).
// The edge set comes from the plan's bitmaps (plan.hip): mask for the rows, maskT for the columns.
// Numerics and operand layouts: dfgnn_dense.hpp (fp16 hi / lo halves under power-of-two scales, fp32-equivalent).
// Replaces, for such ranges, fused_gtconv_backward.cu:40-191 (single-head only there).
#pragma once
#include "dfgnn_dense.hpp"
#include "dfgnn_dense_heads.hpp"
#include "dfgnn_dense_stamp.hpp"

namespace dfgnn {

template <int FR, int NP>
__device__ __forceinline__ void dense_bwd_heads2_body(float *lds, const Csr &g, int n0, int n,
                                                      const float *__restrict__ Q, const float *__restrict__ K,
                                                      const float *__restrict__ V, const float *__restrict__ dO,
                                                      const float *__restrict__ stat_max,
                                                      const float *__restrict__ stat_sum, float *__restrict__ dQ,
                                                      float *__restrict__ dK, float *__restrict__ dV) {
  static_assert(FR == 16 || FR == 32 || FR == 64, "head widths with a 64-column group form");
  static_assert(NP == 128 || NP == 160, "ranges of up to 128 / 160 nodes");
  constexpr int FW = kHeadsGroupWidth, G = FW / FR, U = NP / 16, NX = U - kDenseWaves;  // NX: strips past the eighth
  constexpr int KTH = FR == 64 ? 2 : 1, FTH = FR / 16, MW = (U + 1) / 2;
  constexpr bool kExtra = NX > 0;
  using D = DenseCfg<FW>;
  constexpr int RS = D::RS, KT = D::KT;
  constexpr float kLog2e = 1.4426950408889634f;
  static_assert(4 * NP * RS * 2 + (3 * G * NP + 2 * kDenseWaves) * 4 <= kLdsBytes, "LDS");
  const int ngroups = g.h * FR / FW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int nstrip = (n + 15) >> 4;
  h16 *ahi = reinterpret_cast<h16 *>(lds), *alo = ahi + (size_t)NP * RS;  // image A: K (row pass), Q (col pass)
  h16 *bhi = alo + (size_t)NP * RS, *blo = bhi + (size_t)NP * RS;        // image B: V (row pass), dO (col pass)
  float *tarr = reinterpret_cast<float *>(blo + (size_t)NP * RS);        // [G][NP] t_i of the group's heads
  float *smxl = tarr + G * NP;                                           // [G][NP] m_i log2(e)
  float *sinvl = smxl + G * NP;                                          // [G][NP] 1 / l_i
  float *smax = sinvl + G * NP;                                          // [2][8] per-wave maxima of the two images
  DFGNN_LDS_AT(lds, (unsigned)(reinterpret_cast<char *>(smax + 2 * kDenseWaves) - reinterpret_cast<char *>(lds)));  // the carve-up fits
  const size_t hf = (size_t)g.h * FR;
  const float *Qb = Q + (size_t)n0 * hf, *Kb = K + (size_t)n0 * hf, *Vb = V + (size_t)n0 * hf, *dOb = dO + (size_t)n0 * hf;
  float *dQb = dQ + (size_t)n0 * hf, *dKb = dK + (size_t)n0 * hf, *dVb = dV + (size_t)n0 * hf;
  // this wave's extra unit: strip 8 + x / G, head x % G of every group, for x = wave < NX G
  const int xstrip = kDenseWaves + wave / G, xq = wave % G;
  const bool has_x = kExtra && wave < NX * G && xstrip < nstrip;

  DFGNN_DSTAMP(0)
  // ---- prologue: the bitmaps of this lane's rows and columns (small, first), the first pair of images -------------------
  unsigned mrow[MW], mcol[MW], mrow_x[kExtra ? MW : 1], mcol_x[kExtra ? MW : 1];
  {
    const LaneIds L = lane_ids();
    const int i = wave * 16 + L.mi;
    const size_t node = (size_t)(n0 + min(i, n - 1)) * kPlanMaskWords;
#pragma unroll
    for (int w = 0; w < MW; ++w) {
      const unsigned a = ld32(g.mask + node, (unsigned)w), b = ld32(g.maskT + node, (unsigned)w);
      mrow[w] = (i < n) ? a : 0u;
      mcol[w] = (i < n) ? b : 0u;
    }
    if constexpr (kExtra) {
      const int ix = xstrip * 16 + L.mi;
      const size_t nodex = (size_t)(n0 + min(ix, n - 1)) * kPlanMaskWords;
#pragma unroll
      for (int w = 0; w < MW; ++w) {
        const unsigned a = ld32(g.mask + nodex, (unsigned)w), b = ld32(g.maskT + nodex, (unsigned)w);
        mrow_x[w] = (has_x && ix < n) ? a : 0u;
        mcol_x[w] = (has_x && ix < n) ? b : 0u;
      }
    }
  }
  DenseStageRegs<FW, NP> stA, stB;
  dense_stage_load<FW, NP>(stA, Kb, hf, 0, n);
  dense_stage_load<FW, NP>(stB, Vb, hf, 0, n);
  {
    const int tid = opaque_tid();
    for (int k = tid; k < G * NP; k += kDenseThreads) tarr[k] = 0.f;  // (rows past the last strip are never written)
  }
  // the row statistics of the group's heads -> LDS (one node per thread; past the range: p = 2^0 x 0)
  float st_m[G], st_s[G];
  auto stats_fetch = [&](int gq) {
    const int tid = opaque_tid();
    const size_t at = (size_t)(n0 + min(tid, n - 1)) * g.h + (size_t)gq * G;
#pragma unroll
    for (int q = 0; q < G; ++q) {
      st_m[q] = stat_max[at + q];
      st_s[q] = stat_sum[at + q];
    }
  };
  auto stats_store = [&]() {
    const int tid = opaque_tid();
    if (tid < NP) {
#pragma unroll
      for (int q = 0; q < G; ++q) {
        const bool ok = tid < n && st_s[q] != 0.f;
        smxl[q * NP + tid] = ok ? st_m[q] * kLog2e : 0.f;
        sinvl[q * NP + tid] = ok ? 1.f / st_s[q] : 0.f;
      }
    }
  };
  stats_fetch(0);
  // The register operands of a pass: this lane's 8-float pieces of its strip's rows of two matrices X, Y (the 64 columns
  // of the group), and of the extra unit's rows (its head's k-steps only); fetched raw, converted ONCE per pass to fp16
  // hi / lo fragments under the strip's own power-of-two scale (per matrix).
  float4 xa[KT], xb[KT], ya[KT], yb[KT];
  float4 exa[kExtra ? KTH : 1], exb[kExtra ? KTH : 1], eya[kExtra ? KTH : 1], eyb[kExtra ? KTH : 1];
  auto rows_fetch = [&](const float *X, const float *Y) {
    const LaneIds L = lane_ids();
    const unsigned off = (unsigned)min(wave * 16 + L.mi, n - 1) * (unsigned)hf + 8u * L.mq;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      xa[t] = ld32_f4(X, off + 32 * t);
      xb[t] = ld32_f4(X, off + 32 * t + 4);
      ya[t] = ld32_f4(Y, off + 32 * t);
      yb[t] = ld32_f4(Y, off + 32 * t + 4);
    }
    if constexpr (kExtra) {
      const unsigned offx = (unsigned)min(xstrip * 16 + L.mi, n - 1) * (unsigned)hf + 8u * L.mq + 32u * ((xq * FR) / 32);
#pragma unroll
      for (int t = 0; t < KTH; ++t) {
        exa[t] = ld32_f4(X, offx + 32 * t);
        exb[t] = ld32_f4(X, offx + 32 * t + 4);
        eya[t] = ld32_f4(Y, offx + 32 * t);
        eyb[t] = ld32_f4(Y, offx + 32 * t + 4);
      }
    }
  };
  hx8 xh[KT], xl[KT], yh[KT], yl[KT];
  hx8 exh[kExtra ? KTH : 1], exl[kExtra ? KTH : 1], eyh[kExtra ? KTH : 1], eyl[kExtra ? KTH : 1];
  float xinv = 1.f, yinv = 1.f, exinv = 1.f, eyinv = 1.f;
  auto rows_convert = [&]() {
    const LaneIds L = lane_ids();
    {
      const bool valid = wave * 16 + L.mi < n;
      float mx = 0.f, my = 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        if (!valid) xa[t] = xb[t] = ya[t] = yb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        mx = fmaxf(mx, absmax8(xa[t], xb[t]));
        my = fmaxf(my, absmax8(ya[t], yb[t]));
      }
      const Pow2Scale sx = pow2_scale(wave_max(mx)), sy = pow2_scale(wave_max(my));
      xinv = sx.inv;
      yinv = sy.inv;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        split_hx8(xa[t], xb[t], sx.s, xh[t], xl[t]);
        split_hx8(ya[t], yb[t], sy.s, yh[t], yl[t]);
      }
    }
    if constexpr (kExtra) {
      // (16-wide heads share a 32-deep k-step: the other head's half is zeroed here, once)
      const bool valid = has_x && xstrip * 16 + L.mi < n && (FR >= 32 || (L.mq >> 1) == (xq & 1));
      float mx = 0.f, my = 0.f;
#pragma unroll
      for (int t = 0; t < KTH; ++t) {
        if (!valid) exa[t] = exb[t] = eya[t] = eyb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        mx = fmaxf(mx, absmax8(exa[t], exb[t]));
        my = fmaxf(my, absmax8(eya[t], eyb[t]));
      }
      const Pow2Scale sx = pow2_scale(wave_max(mx)), sy = pow2_scale(wave_max(my));
      exinv = sx.inv;
      eyinv = sy.inv;
#pragma unroll
      for (int t = 0; t < KTH; ++t) {
        split_hx8(exa[t], exb[t], sx.s, exh[t], exl[t]);
        split_hx8(eya[t], eyb[t], sy.s, eyh[t], eyl[t]);
      }
    }
  };
  // head q's k-steps of the converted rows of the wave's own strip (q is a run-time value: selects, no indexing)
  auto head_operand = [&](const hx8 (&h)[KT], const hx8 (&l)[KT], int q, hx8 (&oh)[KTH], hx8 (&ol)[KTH]) {
    const LaneIds L = lane_ids();
    if constexpr (FR == 64) {  // G = 1: the whole group
#pragma unroll
      for (int t = 0; t < KTH; ++t) { oh[t] = h[t]; ol[t] = l[t]; }
    } else {
      const bool second = (q * FR) / 32 != 0;            // which k-step holds the head
      const bool mine = FR >= 32 || (L.mq >> 1) == (q & 1);  // 16-wide heads: which half of it
      const hx8 z = {};
      oh[0] = mine ? (second ? h[1] : h[0]) : z;
      ol[0] = mine ? (second ? l[1] : l[0]) : z;
    }
  };
  // one D^T tile (image rows 16 u ..) of an image against a register row operand, the k-steps of the head that starts at
  // image column c0 (= q FR)
  auto rows_tile = [&](const h16 *ihi, const h16 *ilo, int u, int c0, const hx8 (&oh)[KTH], const hx8 (&ol)[KTH], const LaneIds &L) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int off = (16 * u + L.mi) * RS + 8 * L.mq + (c0 & ~31);
#pragma unroll
    for (int t = 0; t < KTH; ++t) {
      const hx8 ah = *reinterpret_cast<const hx8 *>(ihi + off + 32 * t), al = *reinterpret_cast<const hx8 *>(ilo + off + 32 * t);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, oh[t], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ol[t], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, oh[t], acc, 0, 0, 0);
    }
    return acc;
  };
  // acc[k] += (the head's feature tiles of an image, rows 32 jb ..)^T . Y for ONE 32-deep k-block, Y = the pair (y0, y1) of
  // D^T tiles 2 jb, 2 jb + 1 (accumulators, permuted k order): the "P V" form of the forward
  auto cols_kblock = [&](f32x4 (&acc)[FTH], f32x4 (&aux)[2], const h16 *ihi, const h16 *ilo, int jb, int c0, const f32x4 &y0,
                         const f32x4 &y1, float yscale, const LaneIds &L) {
    hx8 fh, fl;
    dense_split8(y0, y1, yscale, fh, fl);
    dense_kblock_mma_n<FTH>(acc, aux, ihi, ilo, (32 * jb + 4 * L.mq + L.tq) * RS + 4 * L.tp + c0, 16 * RS, fh, fl);
  };
  // Edge masking without compares: the bitmap words of a row / column are shifted right by 4 mq once per unit (then bit
  // 16 (u & 1) + r of word u / 2 says whether pair r of tile u is an edge); per pair one sign-extending bit-field
  // extract gives 0 / ~0, which is ANDed onto the value (a non-edge's exponential may be anything, even inf).
  auto shift_words = [&](const unsigned (&w)[MW], unsigned (&o)[MW]) {
    const LaneIds L = lane_ids();
#pragma unroll
    for (int k = 0; k < MW; ++k) o[k] = w[k] >> (4 * L.mq);
  };
  auto edge_and = [&](float v, const unsigned (&w)[MW], int u, int r) -> float {
    const int m = __builtin_amdgcn_sbfe((int)w[u / 2], 16 * (u & 1) + r, 1);
    return __int_as_float(__float_as_int(v) & m);
  };
  // A running power-of-two scale for an operand that is produced k-block by k-block (dS): it only ever shrinks; the
  // accumulators, kept in units of 1 / scale, are multiplied by the (exact) ratio when it does.
  auto running_scale = [&](float &s_cur, float &inv_cur, float amax, f32x4 (&acc)[FTH], f32x4 (&aux)[2]) {
    const Pow2Scale need = pow2_scale(wave_max(amax));
    if (need.s < s_cur) {  // (wave-uniform)
      const float ratio = inv_cur * need.s;
#pragma unroll
      for (int k = 0; k < FTH; ++k) acc[k] *= ratio;
      aux[0] *= ratio;
      aux[1] *= ratio;
      s_cur = need.s;
      inv_cur = need.inv;
    }
  };
  constexpr float kScaleTop = 1.7014118e38f;  // 2^127: above every scale pow2_scale returns
  // FTH tiles of one head of a 16-row strip -> global rows (whole lines when the head is at least 32 features wide)
  auto head_store = [&](const f32x4 (&acc)[FTH], float scale, float *base, int row, const LaneIds &L) {
    if constexpr (FTH % 2 == 0) {
      dense_store_rows<FTH>(acc, scale, base, (unsigned)hf, row, n, L);
    } else if (row < n) {
      dense_store_acc<FTH>(acc, scale, base, (unsigned)row * (unsigned)hf + 4u * L.mq, false);
    }
  };

  // ---- one unit of the row pass: strip (rows i), head at group column c0; x = its Q rows, y = its dO rows ---------------------
  auto row_unit = [&](int strip, int q, const hx8 (&qh)[KTH], const hx8 (&ql)[KTH], const hx8 (&dh)[KTH], const hx8 (&dl)[KTH],
                      float c2, float dpc, float kinv, const unsigned (&mw)[MW], float *dQg) {
    const LaneIds L = lane_ids();
    const int i = strip * 16 + L.mi, c0 = q * FR;
    const float b2 = smxl[q * NP + i], sinv = sinvl[q * NP + i];
    unsigned ws[MW];
    shift_words(mw, ws);
    // sweep 1: P (kept), t_i = sum_j P dP.  Every tile of the padded range, no branches (the images are zero past the
    // range and the bitmaps have no bits there): straight-line code lets the MFMA chains of one tile run under the vector
    // work of another.
    // The exponentials are normalised by THEIR OWN row sum (the forward's l_i belongs to the forward's rounding of S: with
    // logits of +-100 the two differ by 1e-5 relative, and sum_j P_ij = 1 is what the softmax Jacobian assumes); the col
    // pass gets 1 / l from here as well.  The forward's m_i only has to be near the maximum.
    f32x4 P[U];
    float t = 0.f, l = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4 sacc = rows_tile(ahi, alo, u, c0, qh, ql, L);
      const f32x4 dacc = rows_tile(bhi, blo, u, c0, dh, dl, L);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = edge_and(__builtin_amdgcn_exp2f(fmaf(sacc[r], c2, -b2)), ws, u, r);
        P[u][r] = p;
        l += p;
        t = fmaf(p, dacc[r] * dpc, t);
      }
    }
    l = xor16_32_sum(l);  // a row lives on 4 lanes of this wave
    const float linv = (sinv != 0.f && l > 0.f) ? 1.f / l : 0.f;
    t = xor16_32_sum(t) * linv;
#pragma unroll
    for (int u = 0; u < U; ++u) P[u] *= linv;
    if (L.mq == 0) {
      tarr[q * NP + i] = t;
      sinvl[q * NP + i] = linv;
    }
    // sweep 2: dP again, dS = P (dP - t) k-block by k-block -> dQ^T = K^T dS^T
    f32x4 acc[FTH], aux[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < FTH; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_cur = kScaleTop, inv_cur = 0.f;
#pragma unroll
    for (int jb = 0; jb < NP / 32; ++jb) {
      f32x4 ds0 = rows_tile(bhi, blo, 2 * jb, c0, dh, dl, L), ds1 = rows_tile(bhi, blo, 2 * jb + 1, c0, dh, dl, L);
      float amax = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ds0[r] = P[2 * jb][r] * (ds0[r] * dpc - t);
        ds1[r] = P[2 * jb + 1][r] * (ds1[r] * dpc - t);
        amax = fmaxf(amax, fmaxf(fabsf(ds0[r]), fabsf(ds1[r])));
      }
      running_scale(s_cur, inv_cur, amax, acc, aux);
      cols_kblock(acc, aux, ahi, alo, jb, c0, ds0, ds1, s_cur, L);
    }
    if constexpr (FTH == 1) acc[0] += aux[0] + aux[1];
    head_store(acc, kinv * inv_cur, dQg + c0, i, L);
  };
  // ---- one unit of the col pass: strip (columns j), head at c0; x = its K rows, y = its V rows; lane (mi = column j, mq),
  //      register r of tile u <-> row 16 u + 4 mq + r --------------------------------------------------------------------------
  auto col_unit = [&](int strip, int q, const hx8 (&kh)[KTH], const hx8 (&kl)[KTH], const hx8 (&vh)[KTH], const hx8 (&vl)[KTH],
                      float c2, float dpc, float qinv_img, float doinv_img, const unsigned (&mw)[MW], float *dKg, float *dVg) {
    const LaneIds L = lane_ids();
    const int j = strip * 16 + L.mi, c0 = q * FR;
    f32x4 av[FTH], auxv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 ak[FTH], auxk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < FTH; ++k) av[k] = ak[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_cur = kScaleTop, inv_cur = 0.f;
    unsigned ws[MW];
    shift_words(mw, ws);
#pragma unroll
    for (int ib = 0; ib < NP / 32; ++ib) {  // (every k-block of the padded range, no branches: see row_unit)
      f32x4 pt[2], ds[2];
      float amax = 0.f;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int u = 2 * ib + h2;
        const f32x4 sacc = rows_tile(ahi, alo, u, c0, kh, kl, L);
        const f32x4 dacc = rows_tile(bhi, blo, u, c0, vh, vl, L);
        const float4 b2 = *reinterpret_cast<const float4 *>(smxl + q * NP + 16 * u + 4 * L.mq);
        const float4 si = *reinterpret_cast<const float4 *>(sinvl + q * NP + 16 * u + 4 * L.mq);
        const float4 tt = *reinterpret_cast<const float4 *>(tarr + q * NP + 16 * u + 4 * L.mq);
        const float b2v[4] = {b2.x, b2.y, b2.z, b2.w}, siv[4] = {si.x, si.y, si.z, si.w}, ttv[4] = {tt.x, tt.y, tt.z, tt.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = edge_and(__builtin_amdgcn_exp2f(fmaf(sacc[r], c2, -b2v[r])) * siv[r], ws, u, r);
          pt[h2][r] = p;
          ds[h2][r] = p * (dacc[r] * dpc - ttv[r]);
          amax = fmaxf(amax, fabsf(ds[h2][r]));
        }
      }
      cols_kblock(av, auxv, bhi, blo, ib, c0, pt[0], pt[1], kUnitScale, L);  // dV^T += dO^T P
      running_scale(s_cur, inv_cur, amax, ak, auxk);
      cols_kblock(ak, auxk, ahi, alo, ib, c0, ds[0], ds[1], s_cur, L);        // dK^T += Q^T dS
    }
    if constexpr (FTH == 1) {
      av[0] += auxv[0] + auxv[1];
      ak[0] += auxk[0] + auxk[1];
    }
    head_store(av, doinv_img * kUnitScaleInv, dVg + c0, j, L);
    head_store(ak, qinv_img * inv_cur, dKg + c0, j, L);
  };

  for (int gq = 0; gq < ngroups; ++gq) {  // ---- one group of 64 feature columns (G heads) per trip ----------------------
    const float *Qg = Qb + gq * FW, *Kg = Kb + gq * FW, *Vg = Vb + gq * FW, *dOg = dOb + gq * FW;
    // images: K -> A, V -> B; the group's statistics -> LDS
    wg_max_post(smax, dense_stage_absmax<FW, NP>(stA));
    wg_max_post(smax + kDenseWaves, dense_stage_absmax<FW, NP>(stB));
    lds_barrier();  // B0: every wave is done with the previous group's col pass
    const Pow2Scale ksc = pow2_scale(wg_max_read(smax)), vsc = pow2_scale(wg_max_read(smax + kDenseWaves));
    // memory returns in order: the row pass's register operands are requested first and converted (= waited for) ahead of
    // the barrier, the col pass's images behind them and left in flight
    rows_fetch(Qg, dOg);
    dense_stage_store<FW, NP>(stA, ahi, alo, ksc.s);
    dense_stage_store<FW, NP>(stB, bhi, blo, vsc.s);
    stats_store();
    rows_convert();
    dense_stage_load<FW, NP>(stA, Qg, hf, 0, n);
    dense_stage_load<FW, NP>(stB, dOg, hf, 0, n);
    lds_barrier();  // B1
    if (gq == 0) { DFGNN_DSTAMP(1) }

    // ---- row pass ----------------------------------------------------------------------------------------------------------
    if (wave < nstrip) {
#pragma unroll 1
      for (int q = 0; q < G; ++q) {
        hx8 qh[KTH], ql[KTH], dh[KTH], dl[KTH];
        head_operand(xh, xl, q, qh, ql);
        head_operand(yh, yl, q, dh, dl);
        row_unit(wave, q, qh, ql, dh, dl, (ksc.inv * xinv) * kLog2e, vsc.inv * yinv, ksc.inv, mrow, dQb + gq * FW);
      }
    }
    if constexpr (kExtra) {
      if (has_x) row_unit(xstrip, xq, exh, exl, eyh, eyl, (ksc.inv * exinv) * kLog2e, vsc.inv * eyinv, ksc.inv, mrow_x, dQb + gq * FW);
    }
    if (gq == 0) { DFGNN_DSTAMP(2) }
    wg_max_post(smax, dense_stage_absmax<FW, NP>(stA));                // Q
    wg_max_post(smax + kDenseWaves, dense_stage_absmax<FW, NP>(stB));  // dO
    lds_barrier();  // B2: every wave is done with the K / V images; the t_i are posted
    const Pow2Scale qsc = pow2_scale(wg_max_read(smax)), dosc = pow2_scale(wg_max_read(smax + kDenseWaves));
    rows_fetch(Kg, Vg);
    dense_stage_store<FW, NP>(stA, ahi, alo, qsc.s);
    dense_stage_store<FW, NP>(stB, bhi, blo, dosc.s);
    rows_convert();
    {  // the next group's K and V images and statistics (after the last group: one clamped row each, never stored)
      const bool more = gq + 1 < ngroups;
      const int gn = more ? gq + 1 : gq;
      stats_fetch(gn);
      dense_stage_load<FW, NP>(stA, Kb + gn * FW, hf, 0, more ? n : 1);
      dense_stage_load<FW, NP>(stB, Vb + gn * FW, hf, 0, more ? n : 1);
    }
    lds_barrier();  // B3
    if (gq == 0) { DFGNN_DSTAMP(3) }

    // ---- col pass ----------------------------------------------------------------------------------------------------------
    if (wave < nstrip) {
#pragma unroll 1
      for (int q = 0; q < G; ++q) {
        hx8 kh[KTH], kl[KTH], vh[KTH], vl[KTH];
        head_operand(xh, xl, q, kh, kl);
        head_operand(yh, yl, q, vh, vl);
        col_unit(wave, q, kh, kl, vh, vl, (qsc.inv * xinv) * kLog2e, dosc.inv * yinv, qsc.inv, dosc.inv, mcol, dKb + gq * FW,
                 dVb + gq * FW);
      }
    }
    if constexpr (kExtra) {
      if (has_x)
        col_unit(xstrip, xq, exh, exl, eyh, eyl, (qsc.inv * exinv) * kLog2e, dosc.inv * eyinv, qsc.inv, dosc.inv, mcol_x,
                 dKb + gq * FW, dVb + gq * FW);
    }
    if (gq == 0) { DFGNN_DSTAMP(4) }
  }
  DFGNN_DSTAMP(5)
}

}  // namespace dfgnn
