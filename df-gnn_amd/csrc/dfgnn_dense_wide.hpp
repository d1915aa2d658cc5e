// dfgnn_dense_wide.hpp -- matrix-core GT backward for dense ranges of up to 160 nodes, in ONE row block.
//
// A third of the ranges of the headline workload (PATTERN: 119 +- 21 nodes per graph) have more than 128 nodes.  The
// general backward body (gt_dense.hip: dense_bwd_body<160, 1>) takes them as two row blocks of <= 80 rows, because a
// 160-row P / dS tile (105 KB) and a 160 x 128 feature image (90 KB) do not fit the LDS together; every row block
// then stages the full V and K images again and the second one read-modify-writes dK / dV: 2.4 x the time of a
// 128-node range for 1.4 x the bytes.  This body keeps the whole NP x NP tile resident (NP = 128 or 160) and shrinks
// the image instead: feature matrices are staged 64 columns at a time, eight image phases
//     dO.h -> dV.h (+ the strips' dO operands) ; V.h -> dP += ...          for the feature halves h
//     dS ; K.h -> dQ.h ; Q.h -> dK.h                                        for the feature halves h
// every operand is read from HBM once, nothing is accumulated in global memory.  The strips are dealt to the 8 waves
// round robin (NP = 160: waves 0 and 1 own two): the row products (dP, dS, dQ) run strip after strip, the column
// products (dV, dK) are dealt out in 16 x 16 output tiles, so only the former are unbalanced.
// The half images are fetched through a RING of R register sets, R image phases ahead: what bounds these kernels is the
// number of bytes a CU has in flight (a CU's share of the HBM rate times the loaded latency is ~75 KB; one half image
// is 30 - 40 KB), so a single image in flight leaves every phase waiting for its data.
// P is scattered straight into the tile as fp16 hi | lo halves (scale 2^14) and read back from there (hi + lo equals P
// to 2^-24); dS replaces it in place.  Numerics, layouts and helpers: dfgnn_dense.hpp.
#pragma once
#include "dfgnn_dense.hpp"
#include "dfgnn_dense_stamp.hpp"

namespace dfgnn {

#ifdef DFGNN_STAMPS
extern __device__ unsigned long long *dfgnn_dense_stamps;
#define DFGNN_WSTAMP(k)                                                                             \
  if (threadIdx.x == 0 && dfgnn_dense_stamps)                                                       \
    dfgnn_dense_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define DFGNN_WSTAMP(k)
#endif

// RECOMP (backward of the statistics-saving pair, gt_dense_stats.hip): no attn_edge.  NH more image phases in front --
// K.h -> S += Q.h K.h^T, the strips' Q rows straight from memory as register operands -- then P = 2^(S c - max c) / sum
// on the edges of the plan's bitmap (g.mask) with the forward's row statistics (stat_max, stat_sum: [m, h]), written to
// the tile by its strips (every row of the tile: nothing is cleared, nothing scattered).
// WEIGHTED (with RECOMP; g.wdense = the plan's dense edge values): logits S val, and dS val for dQ / dK (see dense_bwd_rc2_body).
template <int FR, int NP, int R, int FWMAX = 64, bool RECOMP = false, bool WEIGHTED = false>
__device__ __forceinline__ void dense_bwd_wide_body(float *lds, const Csr &g, int n0, int n, int e0, int ne, int head,
                                                    const float *__restrict__ Q, const float *__restrict__ K,
                                                    const float *__restrict__ V, const float *__restrict__ attn_edge,
                                                    const float *__restrict__ dO, float *__restrict__ dQ,
                                                    float *__restrict__ dK, float *__restrict__ dV,
                                                    const float *__restrict__ stat_max = nullptr,
                                                    const float *__restrict__ stat_sum = nullptr) {
  constexpr int F = FR < 32 ? 32 : FR;      // layout width (narrower heads run zero-padded, see dense_fwd_body)
  constexpr int FW = F < FWMAX ? F : FWMAX;  // image width: features staged at a time
  constexpr int NH = F / FW;                // feature halves
  using D = DenseCfg<FW>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT;
  constexpr int U = NP / 16, TS = NP + 8, TB = 2 * TS, NS = (U + kDenseWaves - 1) / kDenseWaves, PRE = kDensePre;
  constexpr int QB = RECOMP ? NH : 0;  // images ahead of dO.0 (RECOMP: K.0 [K.1], for S)
  constexpr int NQ = 4 * NH + QB;      // images in all: dO.0 V.0 [dO.1 V.1] K.0 Q.0 [K.1 Q.1]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int nstrip = (n + 15) >> 4;
  h16 *ihi = reinterpret_cast<h16 *>(lds), *ilo = ihi + (size_t)NP * RS;
  float *T = reinterpret_cast<float *>(ilo + (size_t)NP * RS);
  h16 *Tb = reinterpret_cast<h16 *>(T);
  float *smax = T + NP * TS;  // [8] image maxima, [8] dS maxima
  DFGNN_LDS_AT(lds, (unsigned)(reinterpret_cast<char *>(smax + 2 * kDenseWaves) - reinterpret_cast<char *>(lds)));  // the carve-up fits
  const size_t hf = (size_t)g.h * FR, hoff = (size_t)head * FR;
  const float *Qb = Q + (size_t)n0 * hf + hoff, *Kb = K + (size_t)n0 * hf + hoff, *Vb = V + (size_t)n0 * hf + hoff,
              *dOb = dO + (size_t)n0 * hf + hoff;
  float *dQb = dQ + (size_t)n0 * hf + hoff, *dKb = dK + (size_t)n0 * hf + hoff, *dVb = dV + (size_t)n0 * hf + hoff;
  const float *attn_h = RECOMP ? nullptr : attn_edge + (size_t)head * g.nnz + e0;
  // real feature count of half h (padded widths: FR < 32 -> one half of FR features on the 32-wide layout)
  constexpr int fr = FR < FW ? FR : FW;

  DFGNN_WSTAMP(0)
  // ---- prologue: edges and the first image are requested before anything else ------------------------------------
  unsigned pc[PRE];  // packed (row, column) within the range (plan.hip: coords)
  float pa[PRE];
  if constexpr (!RECOMP) {
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const unsigned e = (unsigned)min(tid + k * kDenseThreads, max(ne, 1) - 1);
      pc[k] = ld32_once(g.coords + e0, e);
      pa[k] = ld32_once(attn_h, e);
    }
  }
  DenseStageRegs<FW, NP> st[R];  // image q travels in st[q % R]
  Pow2Scale isc{1.f, 1.f};
  auto image_src = [&](int qq) -> const float * {  // (qq is a compile-time constant wherever this is called)
    if (qq < QB) return Kb + qq * FW;
    const int q = qq - QB;
    const int h = (q % (2 * NH)) / 2;
    const float *base = q < 2 * NH ? ((q & 1) ? Vb : dOb) : ((q & 1) ? Qb : Kb);
    return base + h * FW;
  };
  auto image_fetch = [&](int q) {
    // (dO -- the even images of the first 2 NH behind the K images of RECOMP -- is read once by the attn_edge form)
    const bool once = !RECOMP && q >= QB && q - QB < 2 * NH && ((q - QB) & 1) == 0;
    if (q < NQ) dense_stage_load<FW, NP>(st[q % R], image_src(q), hf, 0, n, fr, once);
  };
  auto image_post = [&](int q) { wg_max_post(smax, dense_stage_absmax<FW, NP>(st[q % R])); };
  auto image_store = [&](int q) {  // ... and the register set is refilled with the image R phases on
    isc = pow2_scale(wg_max_read(smax));
    dense_stage_store<FW, NP>(st[q % R], ihi, ilo, isc.s, fr);
    if (q == 1) { DFGNN_WSTAMP(13) }
    image_fetch(q + R);
  };
#pragma unroll
  for (int q = 0; q < R; ++q) image_fetch(q);
  if constexpr (RECOMP) {
    // ---- P recomputed: S = Q K^T over the feature halves, then the masked, normalised exponentials -> the tile ---------
    float4 qa[NS][KT], qb[NS][KT];  // this lane's pieces of its strips' Q rows, one feature half, raw
    auto q_fetch = [&](int h) {
      const LaneIds L = lane_ids();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const unsigned off = (unsigned)min((wave + kDenseWaves * s) * 16 + L.mi, n - 1) * (unsigned)hf + (unsigned)(h * FW);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const unsigned c = (fr >= FW || 32 * t + 8 * L.mq < fr) ? 32u * t + 8u * L.mq : 0u;  // (past fr: zeroed below)
          qa[s][t] = ld32_f4(Qb, off + c);
          qb[s][t] = ld32_f4(Qb, off + c + 4);
        }
      }
    };
    hx8 qh[NS][KT], ql[NS][KT];
    float qinv[NS];
    auto q_convert = [&]() {
      const LaneIds L = lane_ids();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const bool valid = (wave + kDenseWaves * s) * 16 + L.mi < n;
        float qm = 0.f;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          if (!valid || (fr < FW && 32 * t + 8 * L.mq >= fr)) qa[s][t] = qb[s][t] = make_float4(0.f, 0.f, 0.f, 0.f);
          qm = fmaxf(qm, absmax8(qa[s][t], qb[s][t]));
        }
        const Pow2Scale qs = pow2_scale(wave_max(qm));
        qinv[s] = qs.inv;
#pragma unroll
        for (int t = 0; t < KT; ++t) split_hx8(qa[s][t], qb[s][t], qs.s, qh[s][t], ql[s][t]);
      }
    };
    q_fetch(0);
    constexpr int MW = (U + 1) / 2;  // bitmap words of a row
    unsigned mwd[NS][MW];
    float smx[NS], sinv[NS];
    {
      const LaneIds L = lane_ids();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int i = (wave + kDenseWaves * s) * 16 + L.mi;
        const size_t node = (size_t)(n0 + min(i, n - 1));
        const unsigned *mp = g.mask + node * kPlanMaskWords;
#pragma unroll
        for (int w = 0; w < MW; ++w) mwd[s][w] = (i < n) ? ld32(mp, (unsigned)w) : 0u;
        smx[s] = stat_max[node * g.h + head];
        const float ssum = stat_sum[node * g.h + head];
        sinv[s] = (ssum != 0.f) ? 1.f / ssum : 0.f;
      }
    }
    f32x4 S[NS][U];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int u = 0; u < U; ++u) S[s][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      image_post(h);
      lds_barrier();  // (h > 0: the previous K half is free)
      image_store(h);  // K, half h
      q_convert();
      if (h + 1 < NH) q_fetch(h + 1);
      lds_barrier();
      const LaneIds L = lane_ids();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (wave + kDenseWaves * s < nstrip) {
          const float c = isc.inv * qinv[s];
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (16 * u < n) S[s][u] += dense_rows_mma<FW>(ihi, ilo, u, qh[s], ql[s], L) * c;
        }
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int strip = wave + kDenseWaves * s;
      if (strip < U) {  // every row of the tile is written: zeros past the range
        const LaneIds L = lane_ids();
        h16 *trow = Tb + (strip * 16 + L.mi) * TB + 4 * L.mq;
        const float b = smx[s];
        // normalised by their own row sum (see dense_bwd_rc2_body): sum_j P_ij = 1 for the P that is differentiated
        float lsum = 0.f;
        const float *wrow = WEIGHTED ? g.wdense + (size_t)(n0 + min(strip * 16 + L.mi, n - 1)) * kPlanWeightStride + 4 * L.mq : nullptr;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned bits = (mwd[s][u / 2] >> (16 * (u & 1) + 4 * L.mq)) & 0xFu;
          if constexpr (WEIGHTED) {
            const float4 wv = ld32_f4(wrow, 16u * u);
            S[s][u][0] *= wv.x; S[s][u][1] *= wv.y; S[s][u][2] *= wv.z; S[s][u][3] *= wv.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            S[s][u][r] = ((bits >> r) & 1u) ? fast_exp(S[s][u][r] - b) : 0.f;
            lsum += S[s][u][r];
          }
        }
        lsum = xor16_32_sum(lsum);
        const float linv = (sinv[s] != 0.f && lsum > 0.f) ? 1.f / lsum : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          hx4 h4, l4;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = S[s][u][r] * linv;
            const h16 hh = (h16)(p * kUnitScale);
            h4[r] = hh;
            l4[r] = (h16)fmaf(p, kUnitScale, -(float)hh);
          }
          *reinterpret_cast<hx4 *>(trow + 16 * u) = h4;
          *reinterpret_cast<hx4 *>(trow + TS + 16 * u) = l4;
        }
      }
    }
  } else {  // tile := 0, then P of every edge as fp16 hi | lo (scale 2^14)
    const int tid = opaque_tid();
    for (int k = tid; k < NP * TS / 4; k += kDenseThreads) reinterpret_cast<float4 *>(T)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    lds_barrier();
    auto put = [&](unsigned c, float p) {
      const int at = (int)(c >> 8) * TB + (int)(c & 0xFF);
      const h16 hh = (h16)(p * kUnitScale);
      Tb[at] = hh;
      Tb[at + TS] = (h16)fmaf(p, kUnitScale, -(float)hh);
    };
#pragma unroll
    for (int k = 0; k < PRE; ++k)
      if (tid + k * kDenseThreads < ne) put(pc[k], pa[k]);
    constexpr int B = 8;
    for (int base = PRE * kDenseThreads; base < ne; base += B * kDenseThreads) {
      unsigned bc[B];
      float ba[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const unsigned e = (unsigned)min(base + tid + k * kDenseThreads, ne - 1);
        bc[k] = ld32_once(g.coords + e0, e);
        ba[k] = ld32_once(attn_h, e);
      }
#pragma unroll
      for (int k = 0; k < B; ++k)
        if (base + tid + k * kDenseThreads < ne) put(bc[k], ba[k]);
    }
  }
  image_post(QB);
  lds_barrier();
  image_store(QB);  // dO, half 0
  lds_barrier();
  DFGNN_WSTAMP(1)

  // out^T[f][c] = sum_i X[i][f] Y[i][c] (X = the image, Y = the tile) for the output tile (column strip cs, feature
  // tile ft); a whole strip (all FT tiles, the Y fragments shared) when ft < 0
  auto column_unit = [&](float *outb, int cs, int ft, float oscale) {
    const LaneIds L = lane_ids();
    const int j = cs * 16 + L.mi;
    if (ft < 0) {
      f32x4 acc[FT];
#pragma unroll
      for (int t = 0; t < FT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ib = 0; ib < NP / 32; ++ib) {
        if (32 * ib < n) {
          const int yoff = (32 * ib + 4 * L.mq + L.tq) * TB + 16 * cs + 4 * L.tp;
          const hx8 yh = dense_tr_pair(Tb + yoff, 16 * TB);
          const hx8 yl = dense_tr_pair(Tb + yoff + TS, 16 * TB);
          dense_kblock_mma<FW, (FW > 64 ? 8 : 4)>(acc, ihi, ilo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp, 16 * RS, yh, yl);
        }
      }
      if constexpr (FR == F) dense_store_rows<FT>(acc, oscale, outb, (unsigned)hf, j, n, L);
      else if (j < n) dense_store_acc<FT, true>(acc, oscale, outb, (unsigned)j * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
    } else {
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int ib = 0; ib < NP / 32; ++ib) {
        if (32 * ib < n) {
          const int yoff = (32 * ib + 4 * L.mq + L.tq) * TB + 16 * cs + 4 * L.tp;
          const hx8 yh = dense_tr_pair(Tb + yoff, 16 * TB);
          const hx8 yl = dense_tr_pair(Tb + yoff + TS, 16 * TB);
          const int xoff = (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp + 16 * ft;
          const hx8 xh = dense_tr_pair(ihi + xoff, 16 * RS);
          const hx8 xl = dense_tr_pair(ilo + xoff, 16 * RS);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yh, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, yh, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yl, acc[0], 0, 0, 0);
        }
      }
      if (j < n)
        dense_store_acc<1, (FR < F)>(acc, oscale, outb, (unsigned)j * (unsigned)hf + 16u * ft + 4u * L.mq, false, 16 * ft + 4 * L.mq, fr);
    }
  };
  // all column strips of the range: the first 8 whole, the others dealt out tile by tile
  auto column_phase = [&](float *outb, float oscale) {
    if (wave < nstrip) column_unit(outb, wave, -1, oscale);
    for (int unit = wave; unit < (nstrip - kDenseWaves) * FT; unit += kDenseWaves)
      column_unit(outb, kDenseWaves + unit / FT, unit % FT, oscale);
  };

  // ---- dV.h = dO.h^T P ; dP += V.h dO.h^T ----------------------------------------------------------------------------
  f32x4 dP[NS][U];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int u = 0; u < U; ++u) dP[s][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    // image = dO, half h; tile = P
    const float doinv = isc.inv;
    hx8 gh[NS][KT], gl[NS][KT];  // the strips' dO rows: the register operand of dP (same scale as the image)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      const int strip = wave + kDenseWaves * s;
      const int off = (min(strip, U - 1) * 16 + L.mi) * RS + 8 * L.mq;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        gh[s][t] = *reinterpret_cast<const hx8 *>(ihi + off + 32 * t);
        gl[s][t] = *reinterpret_cast<const hx8 *>(ilo + off + 32 * t);
      }
    }
    column_phase(dVb + h * FW, doinv * kUnitScaleInv);
    if (h == 0) { DFGNN_WSTAMP(2) }
    image_post(QB + 2 * h + 1);
    lds_barrier();  // the dO image is free
    if (h == 0) { DFGNN_WSTAMP(12) }
    image_store(QB + 2 * h + 1);  // V, half h
    if (h == 0) { DFGNN_WSTAMP(14) }
    lds_barrier();
    if (h == 0) { DFGNN_WSTAMP(3) }
    {
      const float c = isc.inv * doinv;
      const LaneIds L = lane_ids();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (wave + kDenseWaves * s < nstrip) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (16 * u < n) {
              const f32x4 a = dense_rows_mma<FW>(ihi, ilo, u, gh[s], gl[s], L);
              dP[s][u] += a * c;
            }
          }
        }
      }
    }
    if (h == 0) { DFGNN_WSTAMP(4) }
    if (h + 1 < NH) {
      image_post(QB + 2 * h + 2);
      lds_barrier();  // the V image is free
      image_store(QB + 2 * h + 2);  // dO, half h + 1
      lds_barrier();
    }
  }

  DFGNN_WSTAMP(5)
  // ---- t_i = sum_j P dP ; dS = P (dP - t), in place of dP -------------------------------------------------------------
  float tmax = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int strip = wave + kDenseWaves * s;
    if (strip < nstrip) {
      const LaneIds L = lane_ids();
      const h16 *prow = Tb + (strip * 16 + L.mi) * TB + 4 * L.mq;
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const hx4 ph = *reinterpret_cast<const hx4 *>(prow + 16 * u), pl = *reinterpret_cast<const hx4 *>(prow + TS + 16 * u);
#pragma unroll
        for (int r = 0; r < 4; ++r) t = fmaf(((float)ph[r] + (float)pl[r]) * kUnitScaleInv, dP[s][u][r], t);
      }
      t = xor16_32_sum(t);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const hx4 ph = *reinterpret_cast<const hx4 *>(prow + 16 * u), pl = *reinterpret_cast<const hx4 *>(prow + TS + 16 * u);
        float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);  // WEIGHTED: d logit / d S (P is exactly zero off the edges)
        if constexpr (WEIGHTED)
          wv = ld32_f4(g.wdense + (size_t)(n0 + min(strip * 16 + L.mi, n - 1)) * kPlanWeightStride + 4 * L.mq, 16u * u);
        const float wvr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = ((float)ph[r] + (float)pl[r]) * kUnitScaleInv;
          dP[s][u][r] = p * (dP[s][u][r] - t);
          if constexpr (WEIGHTED) dP[s][u][r] = (p != 0.f) ? dP[s][u][r] * wvr[r] : 0.f;
          tmax = fmaxf(tmax, fabsf(dP[s][u][r]));
        }
      }
    }
  }
  DFGNN_WSTAMP(6)
  wg_max_post(smax + kDenseWaves, tmax);
  image_post(QB + 2 * NH);
  lds_barrier();  // the V image and every strip's P rows are free
  const Pow2Scale ts = pow2_scale(wg_max_read(smax + kDenseWaves));
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int strip = wave + kDenseWaves * s;
    if (strip < nstrip) {
      const LaneIds L = lane_ids();
      h16 *trow = Tb + (strip * 16 + L.mi) * TB + 4 * L.mq;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        hx4 h4, l4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const h16 hh = (h16)(dP[s][u][r] * ts.s);
          h4[r] = hh;
          l4[r] = (h16)fmaf(dP[s][u][r], ts.s, -(float)hh);
        }
        *reinterpret_cast<hx4 *>(trow + 16 * u) = h4;
        *reinterpret_cast<hx4 *>(trow + TS + 16 * u) = l4;
      }
    }
  }
  image_store(QB + 2 * NH);  // K, half 0
  lds_barrier();
  DFGNN_WSTAMP(7)

  // ---- dQ.h = dS K.h ; dK.h = dS^T Q.h ----------------------------------------------------------------------------------
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    // image = K, half h; tile = dS
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int strip = wave + kDenseWaves * s;
      if (strip < nstrip) {
        const LaneIds L = lane_ids();
        f32x4 qacc[FT];
#pragma unroll
        for (int t = 0; t < FT; ++t) qacc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const h16 *srow = Tb + (strip * 16 + L.mi) * TB + 8 * L.mq;
#pragma unroll
        for (int jb = 0; jb < NP / 32; ++jb) {
          if (32 * jb < n) {
            const hx8 sh = *reinterpret_cast<const hx8 *>(srow + 32 * jb);
            const hx8 sl = *reinterpret_cast<const hx8 *>(srow + TS + 32 * jb);
            dense_kblock_mma<FW, (FW > 64 ? 8 : 4)>(qacc, ihi, ilo, (32 * jb + 8 * L.mq + L.tq) * RS + 4 * L.tp, 4 * RS, sh, sl);
          }
        }
        const int i = strip * 16 + L.mi;
        if constexpr (FR == F) dense_store_rows<FT>(qacc, isc.inv * ts.inv, dQb + h * FW, (unsigned)hf, i, n, L);
        else if (i < n)
          dense_store_acc<FT, true>(qacc, isc.inv * ts.inv, dQb + h * FW, (unsigned)i * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
      }
    }
    if (h == 0) { DFGNN_WSTAMP(8) }
    image_post(QB + 2 * NH + 2 * h + 1);
    lds_barrier();  // the K image is free
    image_store(QB + 2 * NH + 2 * h + 1);  // Q, half h
    lds_barrier();
    if (h == 0) { DFGNN_WSTAMP(9) }
    column_phase(dKb + h * FW, isc.inv * ts.inv);
    if (h == 0) { DFGNN_WSTAMP(10) }
    if (h + 1 < NH) {
      image_post(QB + 2 * NH + 2 * h + 2);
      lds_barrier();  // the Q image is free
      image_store(QB + 2 * NH + 2 * h + 2);  // K, half h + 1
      lds_barrier();
    }
  }
  DFGNN_WSTAMP(11)
}

}  // namespace dfgnn
