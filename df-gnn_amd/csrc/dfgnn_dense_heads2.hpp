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
// per pass (256-byte row segments from memory), three barriers per group, none per head; the next pass's images
// travel in registers meanwhile.  The register operands of a pass come straight from memory (32-byte pieces, as the
// forward fetches Q); the other pass stages the same rows as an image a few microseconds later (an L2 hit).
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
  constexpr int FW = kHeadsGroupWidth, G = FW / FR, U = NP / 16, NS = (U + kDenseWaves - 1) / kDenseWaves;
  constexpr int KTH = FR == 64 ? 2 : 1, FTH = FR / 16, MW = (U + 1) / 2;
  using D = DenseCfg<FW>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT;
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
  const size_t hf = (size_t)g.h * FR;
  const float *Qb = Q + (size_t)n0 * hf, *Kb = K + (size_t)n0 * hf, *Vb = V + (size_t)n0 * hf, *dOb = dO + (size_t)n0 * hf;
  float *dQb = dQ + (size_t)n0 * hf, *dKb = dK + (size_t)n0 * hf, *dVb = dV + (size_t)n0 * hf;

  DFGNN_DSTAMP(0)
  // ---- prologue: the first pair of images, the bitmaps of this lane's rows and columns ---------------------------------
  DenseStageRegs<FW, NP> stA, stB;
  dense_stage_load<FW, NP>(stA, Kb, hf, 0, n);
  dense_stage_load<FW, NP>(stB, Vb, hf, 0, n);
  unsigned mrow[NS][MW], mcol[NS][MW];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const LaneIds L = lane_ids();
    const int i = (wave + kDenseWaves * s) * 16 + L.mi;
    const size_t node = (size_t)(n0 + min(i, n - 1)) * kPlanMaskWords;
#pragma unroll
    for (int w = 0; w < MW; ++w) {
      mrow[s][w] = (i < n) ? ld32(g.mask + node, (unsigned)w) : 0u;
      mcol[s][w] = (i < n) ? ld32(g.maskT + node, (unsigned)w) : 0u;
    }
  }
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
  // this lane's 8-float pieces of rows of two matrices (one 64-column group): the register operands of a pass, fetched
  // raw and converted ONCE per strip to fp16 hi / lo fragments under the strip's own power-of-two scale (per matrix)
  float4 xa[KT], xb[KT], ya[KT], yb[KT];
  auto rows_fetch = [&](const float *X, const float *Y, int strip) {
    const LaneIds L = lane_ids();
    const unsigned off = (unsigned)min(strip * 16 + L.mi, n - 1) * (unsigned)hf + 8u * L.mq;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      xa[t] = ld32_f4(X, off + 32 * t);
      xb[t] = ld32_f4(X, off + 32 * t + 4);
      ya[t] = ld32_f4(Y, off + 32 * t);
      yb[t] = ld32_f4(Y, off + 32 * t + 4);
    }
  };
  hx8 xh[KT], xl[KT], yh[KT], yl[KT];
  float xinv = 1.f, yinv = 1.f;
  auto rows_convert = [&](int strip) {
    const LaneIds L = lane_ids();
    const bool valid = strip * 16 + L.mi < n;
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
  };
  // head q's k-steps of the converted rows (16-wide heads share a 32-deep k-step: the other head's half is zeroed)
  auto head_operand = [&](const hx8 (&h)[KT], const hx8 (&l)[KT], int q, hx8 (&oh)[KTH], hx8 (&ol)[KTH]) {
    const LaneIds L = lane_ids();
    const int t0 = (q * FR) / 32;
    const bool mine = FR >= 32 || (L.mq >> 1) == (q & 1);
#pragma unroll
    for (int t = 0; t < KTH; ++t) {
      oh[t] = mine ? h[t0 + t] : hx8{};
      ol[t] = mine ? l[t0 + t] : hx8{};
    }
  };
  // one D^T tile (image rows 16 u ..) of an image against a register row operand, head q's k-steps only
  auto rows_tile = [&](const h16 *ihi, const h16 *ilo, int u, int q, const hx8 (&oh)[KTH], const hx8 (&ol)[KTH], const LaneIds &L) {
    const int t0 = (q * FR) / 32;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KTH; ++t) {
      const int off = (16 * u + L.mi) * RS + 8 * L.mq + 32 * (t0 + t);
      const hx8 ah = *reinterpret_cast<const hx8 *>(ihi + off), al = *reinterpret_cast<const hx8 *>(ilo + off);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, oh[t], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ol[t], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, oh[t], acc, 0, 0, 0);
    }
    return acc;
  };
  // acc[k] += (head q's feature tiles of an image, rows 32 jb ..)^T . Y for ONE 32-deep k-block, Y = the pair (y0, y1) of
  // D^T tiles 2 jb, 2 jb + 1 (accumulators, permuted k order): the "P V" form of the forward
  auto cols_kblock = [&](f32x4 (&acc)[FTH], f32x4 (&aux)[2], const h16 *ihi, const h16 *ilo, int jb, int q, const f32x4 &y0,
                         const f32x4 &y1, float yscale, const LaneIds &L) {
    hx8 fh, fl;
    dense_split8(y0, y1, yscale, fh, fl);
    dense_kblock_mma_n<FTH>(acc, aux, ihi, ilo, (32 * jb + 4 * L.mq + L.tq) * RS + 4 * L.tp + 16 * (q * FTH), 16 * RS, fh, fl);
  };
  // the 4 edge bits of tile u for this lane: bits 16 (u & 1) + 4 mq .. + 3 of word u / 2 of a bitmap row
  auto tile_bits = [&](const unsigned (&w)[MW], int u, int mq) -> unsigned { return (w[u / 2] >> (16 * (u & 1) + 4 * mq)) & 0xFu; };
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

  for (int gq = 0; gq < ngroups; ++gq) {  // ---- one group of 64 feature columns (G heads) per trip ----------------------
    const float *Qg = Qb + gq * FW, *Kg = Kb + gq * FW, *Vg = Vb + gq * FW, *dOg = dOb + gq * FW;
    // images: K -> A, V -> B; the group's statistics -> LDS
    wg_max_post(smax, dense_stage_absmax<FW, NP>(stA));
    wg_max_post(smax + kDenseWaves, dense_stage_absmax<FW, NP>(stB));
    lds_barrier();  // B0: every wave is done with the previous group's col pass
    const Pow2Scale ksc = pow2_scale(wg_max_read(smax)), vsc = pow2_scale(wg_max_read(smax + kDenseWaves));
    dense_stage_store<FW, NP>(stA, ahi, alo, ksc.s);
    dense_stage_store<FW, NP>(stB, bhi, blo, vsc.s);
    stats_store();
    dense_stage_load<FW, NP>(stA, Qg, hf, 0, n);  // the col pass's images travel during the row pass
    dense_stage_load<FW, NP>(stB, dOg, hf, 0, n);
    rows_fetch(Qg, dOg, wave);
    lds_barrier();  // B1
    if (gq == 0) { DFGNN_DSTAMP(1) }

    // ---- row pass: x = Q rows, y = dO rows of the strip ----------------------------------------------------------------------
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int strip = wave + kDenseWaves * s;
      if (s > 0 && strip < nstrip) rows_fetch(Qg, dOg, strip);
      if (strip < nstrip) {
        rows_convert(strip);
        const LaneIds L = lane_ids();
        const int i = strip * 16 + L.mi;
        f32x4 o[FT];
#pragma unroll
        for (int q = 0; q < G; ++q) {
          hx8 qh[KTH], ql[KTH], dh[KTH], dl[KTH];
          head_operand(xh, xl, q, qh, ql);
          head_operand(yh, yl, q, dh, dl);
          const float c2 = (ksc.inv * xinv) * kLog2e, b2 = smxl[q * NP + i], sinv = sinvl[q * NP + i];
          const float dpc = vsc.inv * yinv;
          // sweep 1: P (kept), t_i = sum_j P dP
          f32x4 P[U];
          float t = 0.f;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            P[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (16 * u < n) {
              const f32x4 sacc = rows_tile(ahi, alo, u, q, qh, ql, L);
              const f32x4 dacc = rows_tile(bhi, blo, u, q, dh, dl, L);
              const unsigned bits = tile_bits(mrow[s], u, L.mq);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float p = ((bits >> r) & 1u) ? __builtin_amdgcn_exp2f(fmaf(sacc[r], c2, -b2)) * sinv : 0.f;
                P[u][r] = p;
                t = fmaf(p, dacc[r] * dpc, t);
              }
            }
          }
          t = xor16_32_sum(t);  // a row lives on 4 lanes of this wave
          if (L.mq == 0) tarr[q * NP + i] = t;
          // sweep 2: dP again, dS = P (dP - t) k-block by k-block -> dQ^T = K^T dS^T
          f32x4 acc[FTH], aux[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int k = 0; k < FTH; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          float s_cur = kScaleTop, inv_cur = 0.f;
#pragma unroll
          for (int jb = 0; jb < NP / 32; ++jb) {
            if (32 * jb < n) {
              f32x4 ds0 = rows_tile(bhi, blo, 2 * jb, q, dh, dl, L), ds1 = rows_tile(bhi, blo, 2 * jb + 1, q, dh, dl, L);
              float amax = 0.f;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                ds0[r] = P[2 * jb][r] * (ds0[r] * dpc - t);
                ds1[r] = P[2 * jb + 1][r] * (ds1[r] * dpc - t);
                amax = fmaxf(amax, fmaxf(fabsf(ds0[r]), fabsf(ds1[r])));
              }
              running_scale(s_cur, inv_cur, amax, acc, aux);
              cols_kblock(acc, aux, ahi, alo, jb, q, ds0, ds1, s_cur, L);
            }
          }
          if constexpr (FTH == 1) acc[0] += aux[0] + aux[1];
          const float oscale = ksc.inv * inv_cur;
#pragma unroll
          for (int k = 0; k < FTH; ++k) o[q * FTH + k] = acc[k] * oscale;
        }
        dense_store_rows<FT>(o, 1.f, dQb + gq * FW, (unsigned)hf, i, n, L);
      }
    }
    if (gq == 0) { DFGNN_DSTAMP(2) }
    wg_max_post(smax, dense_stage_absmax<FW, NP>(stA));                // Q
    wg_max_post(smax + kDenseWaves, dense_stage_absmax<FW, NP>(stB));  // dO
    lds_barrier();  // B2: every wave is done with the K / V images; the t_i are posted
    const Pow2Scale qsc = pow2_scale(wg_max_read(smax)), dosc = pow2_scale(wg_max_read(smax + kDenseWaves));
    dense_stage_store<FW, NP>(stA, ahi, alo, qsc.s);
    dense_stage_store<FW, NP>(stB, bhi, blo, dosc.s);
    {  // the next group's K and V images and statistics (after the last group: one clamped row each, never stored)
      const bool more = gq + 1 < ngroups;
      const int gn = more ? gq + 1 : gq;
      dense_stage_load<FW, NP>(stA, Kb + gn * FW, hf, 0, more ? n : 1);
      dense_stage_load<FW, NP>(stB, Vb + gn * FW, hf, 0, more ? n : 1);
      stats_fetch(gn);
    }
    rows_fetch(Kg, Vg, wave);
    lds_barrier();  // B3
    if (gq == 0) { DFGNN_DSTAMP(3) }

    // ---- col pass: x = K rows, y = V rows of the strip; lane (mi = column j, mq), register r of tile u <-> row 16 u + 4 mq + r
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int strip = wave + kDenseWaves * s;
      if (s > 0 && strip < nstrip) rows_fetch(Kg, Vg, strip);
      if (strip < nstrip) {
        rows_convert(strip);
        const LaneIds L = lane_ids();
        const int j = strip * 16 + L.mi;
        f32x4 ok[FT], ov[FT];
#pragma unroll
        for (int q = 0; q < G; ++q) {
          hx8 kh[KTH], kl[KTH], vh[KTH], vl[KTH];
          head_operand(xh, xl, q, kh, kl);
          head_operand(yh, yl, q, vh, vl);
          const float c2 = (qsc.inv * xinv) * kLog2e, dpc = dosc.inv * yinv;
          f32x4 av[FTH], auxv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
          f32x4 ak[FTH], auxk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
          for (int k = 0; k < FTH; ++k) av[k] = ak[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          float s_cur = kScaleTop, inv_cur = 0.f;
#pragma unroll
          for (int ib = 0; ib < NP / 32; ++ib) {
            if (32 * ib < n) {
              f32x4 pt[2], ds[2];
              float amax = 0.f;
#pragma unroll
              for (int h2 = 0; h2 < 2; ++h2) {
                const int u = 2 * ib + h2;
                const f32x4 sacc = rows_tile(ahi, alo, u, q, kh, kl, L);
                const f32x4 dacc = rows_tile(bhi, blo, u, q, vh, vl, L);
                const unsigned bits = tile_bits(mcol[s], u, L.mq);
                const float4 b2 = *reinterpret_cast<const float4 *>(smxl + q * NP + 16 * u + 4 * L.mq);
                const float4 si = *reinterpret_cast<const float4 *>(sinvl + q * NP + 16 * u + 4 * L.mq);
                const float4 tt = *reinterpret_cast<const float4 *>(tarr + q * NP + 16 * u + 4 * L.mq);
                const float b2v[4] = {b2.x, b2.y, b2.z, b2.w}, siv[4] = {si.x, si.y, si.z, si.w}, ttv[4] = {tt.x, tt.y, tt.z, tt.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float p = ((bits >> r) & 1u) ? __builtin_amdgcn_exp2f(fmaf(sacc[r], c2, -b2v[r])) * siv[r] : 0.f;
                  pt[h2][r] = p;
                  ds[h2][r] = p * (dacc[r] * dpc - ttv[r]);
                  amax = fmaxf(amax, fabsf(ds[h2][r]));
                }
              }
              cols_kblock(av, auxv, bhi, blo, ib, q, pt[0], pt[1], kUnitScale, L);  // dV^T += dO^T P
              running_scale(s_cur, inv_cur, amax, ak, auxk);
              cols_kblock(ak, auxk, ahi, alo, ib, q, ds[0], ds[1], s_cur, L);        // dK^T += Q^T dS
            }
          }
          if constexpr (FTH == 1) {
            av[0] += auxv[0] + auxv[1];
            ak[0] += auxk[0] + auxk[1];
          }
          const float vscale = dosc.inv * kUnitScaleInv, kscale = qsc.inv * inv_cur;
#pragma unroll
          for (int k = 0; k < FTH; ++k) {
            ov[q * FTH + k] = av[k] * vscale;
            ok[q * FTH + k] = ak[k] * kscale;
          }
        }
        dense_store_rows<FT>(ov, 1.f, dVb + gq * FW, (unsigned)hf, j, n, L);
        dense_store_rows<FT>(ok, 1.f, dKb + gq * FW, (unsigned)hf, j, n, L);
      }
    }
    if (gq == 0) { DFGNN_DSTAMP(4) }
  }
  DFGNN_DSTAMP(5)
}

}  // namespace dfgnn
