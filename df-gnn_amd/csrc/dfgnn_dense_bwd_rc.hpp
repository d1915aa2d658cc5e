// dfgnn_dense_bwd_rc.hpp -- matrix-core GT backward of the statistics-saving pair for dense ranges of <= 128 nodes:
// P recomputed, four image phases, no edge list.
//
// The attn_edge-based body (dfgnn_dense_bwd.hpp) starts from P: scatter it into the tile, dV^T = dO^T P, then V, K, Q.
// Without attn_edge P comes from S = Q K^T, i.e. from the K image -- which the same body needs again three phases later
// for dQ (its RECOMP form stages K twice: five image phases).  Here the phases are ordered so that every image is staged
// ONCE and dQ needs no tile:
//     V  ->  dP^T = V dO^T                                   (the strip's dO rows: register operands, straight from memory)
//     K  ->  S^T = K Q^T, P, t_i = sum_j P dP, dS = P (dP - t)   (the strip's Q rows likewise)
//            dQ^T = K^T dS^T with the dS accumulators AS the operand (dfgnn_dense.hpp: a D^T strip is the B operand of
//            the product that contracts over its rows -- the forward's P V form), K still resident;  P -> tile
//     dO ->  dV^T = dO^T P     (tile);  then dS -> tile
//     Q  ->  dK^T = Q^T dS     (tile)
// The strips' dO and Q rows are read twice (as 32-byte register pieces first, as an image a few microseconds later: an
// L2 / Infinity Cache hit); HBM sees every operand once.  P_ij = 2^(S_ij c - m_i c) / l_i on the edges of the plan's
// bitmap of row i (g.mask), (m_i, l_i) = the forward's row statistics.  Numerics and layouts: dfgnn_dense.hpp.
// Replaces, for such ranges, fused_gtconv_backward.cu:40-191 (with gt_dense_stats.hip's forward in place of :31-163).
#pragma once
#include "dfgnn_dense.hpp"
#include "dfgnn_dense_stamp.hpp"

namespace dfgnn {

template <int FR>
__device__ __forceinline__ void dense_bwd_rc_body(float *lds, const Csr &g, int n0, int n, int head,
                                                  const float *__restrict__ Q, const float *__restrict__ K,
                                                  const float *__restrict__ V, const float *__restrict__ stat_max,
                                                  const float *__restrict__ stat_sum, const float *__restrict__ dO,
                                                  float *__restrict__ dQ, float *__restrict__ dK,
                                                  float *__restrict__ dV) {
  constexpr int F = FR < 32 ? 32 : FR;  // layout width (narrower heads run zero-padded, see dense_fwd_body)
  constexpr int fr = FR, CW = kDenseChunkRows, U = CW / 16, TS = CW + 8, TB = 2 * TS;
  using D = DenseCfg<F>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT;
  constexpr float kLog2e = 1.4426950408889634f;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // = this wave's strip
  h16 *ihi = reinterpret_cast<h16 *>(lds), *ilo = ihi + (size_t)CW * RS;
  float *T = reinterpret_cast<float *>(ilo + (size_t)CW * RS);
  h16 *Tb = reinterpret_cast<h16 *>(T);   // the tile: CW rows of hi[TS] | lo[TS]
  float *smax = T + CW * TS;              // [8] per-wave maxima of the image being staged, [8] of dS
  const size_t hf = (size_t)g.h * fr, hoff = (size_t)head * fr;
  const float *Qb = Q + (size_t)n0 * hf + hoff, *Kb = K + (size_t)n0 * hf + hoff, *Vb = V + (size_t)n0 * hf + hoff,
              *dOb = dO + (size_t)n0 * hf + hoff;
  float *dQb = dQ + (size_t)n0 * hf + hoff, *dKb = dK + (size_t)n0 * hf + hoff, *dVb = dV + (size_t)n0 * hf + hoff;
  const bool row_wave = wave * 16 < n;

  DFGNN_DSTAMP(0)
  // ---- prologue: the V image, this lane's pieces of its dO row, its row's bitmap and statistics --------------------------
  DenseStageRegs<F, CW> st;
  dense_stage_load<F, CW>(st, Vb, hf, 0, n, fr);
  float4 ra[KT], rb[KT];  // raw pieces of a row (dO, then Q)
  auto row_fetch = [&](const float *X) {
    const LaneIds L = lane_ids();
    const unsigned off = (unsigned)min(wave * 16 + L.mi, n - 1) * (unsigned)hf;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const unsigned c = (FR == F || 32 * t + 8 * L.mq < fr) ? 32u * t + 8u * L.mq : 0u;  // (past fr: zeroed below)
#ifdef DFGNN_RC_NOFRAG  // timing experiment only (wrong results): what the register operands' loads cost
      ra[t] = rb[t] = make_float4(1.f + (float)c, 2.f, 3.f, (float)off);
#else
      ra[t] = ld32_f4(X, off + c);
      rb[t] = ld32_f4(X, off + c + 4);
#endif
    }
  };
  // ... as fp16 hi / lo operand fragments under the strip's own power-of-two scale; returns 1 / scale
  auto row_convert = [&](hx8 (&oh)[KT], hx8 (&ol)[KT]) {
    const LaneIds L = lane_ids();
    const bool valid = wave * 16 + L.mi < n;
    float mx = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (!valid || (FR < F && 32 * t + 8 * L.mq >= fr)) ra[t] = rb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      mx = fmaxf(mx, absmax8(ra[t], rb[t]));
    }
    const Pow2Scale sc = pow2_scale(wave_max(mx));
#pragma unroll
    for (int t = 0; t < KT; ++t) split_hx8(ra[t], rb[t], sc.s, oh[t], ol[t]);
    return sc.inv;
  };
  row_fetch(dOb);
  unsigned mwd[U / 2];
  float smx, sinv;
  {
    const LaneIds L = lane_ids();
    const int i = wave * 16 + L.mi;
    const size_t node = (size_t)(n0 + min(i, n - 1));
    const uint4 w = *reinterpret_cast<const uint4 *>(g.mask + node * kPlanMaskWords);
    const bool valid = i < n;
    mwd[0] = valid ? w.x : 0u; mwd[1] = valid ? w.y : 0u; mwd[2] = valid ? w.z : 0u; mwd[3] = valid ? w.w : 0u;
    smx = stat_max[node * g.h + head] * kLog2e;
    const float ssum = stat_sum[node * g.h + head];
    sinv = (valid && ssum != 0.f) ? 1.f / ssum : 0.f;
  }
  Pow2Scale isc{1.f, 1.f};
  auto image_commit = [&](const float *next) {  // registers -> LDS around the barrier pair; the next image is requested between
    wg_max_post(smax, dense_stage_absmax<F, CW>(st));
    lds_barrier();
    isc = pow2_scale(wg_max_read(smax));
    dense_stage_store<F, CW>(st, ihi, ilo, isc.s, fr);
    if (next) dense_stage_load<F, CW>(st, next, hf, 0, n, fr);
  };
  // this strip's 16 x CW values (times a power-of-two scale) -> its own rows of the tile, as interleaved fp16 hi | lo
  auto strip_to_tile = [&](const f32x4 (&X)[U], float tscale) {
    const LaneIds L = lane_ids();
    h16 *trow = Tb + (wave * 16 + L.mi) * TB + 4 * L.mq;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      hx4 h4, l4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const h16 h = (h16)(X[u][r] * tscale);
        h4[r] = h;
        l4[r] = (h16)fmaf(X[u][r], tscale, -(float)h);
      }
      *reinterpret_cast<hx4 *>(trow + 16 * u) = h4;
      *reinterpret_cast<hx4 *>(trow + TS + 16 * u) = l4;
    }
  };
  // out^T[f][c] = sum_i X[i][f] Y[i][c], X = the image, Y = the tile: wave w takes the column strips 2 (w / 2),
  // 2 (w / 2) + 1 and the feature tiles of half w % 2 (the image fragments are shared by the two strips); narrow widths:
  // a column strip per wave
  auto column_product = [&](float *outb, float oscale) {
    const LaneIds L = lane_ids();
    const int nstrips = (n + 15) >> 4;
    if constexpr (FR == F && FT >= 4) {
      constexpr int NFT = FT / 2;
      const int cs0 = 2 * (wave >> 1), ft0 = NFT * (wave & 1);
      if (cs0 < nstrips) {
        f32x4 acc0[NFT], acc1[NFT];
#pragma unroll
        for (int k = 0; k < NFT; ++k) acc0[k] = acc1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < CW / 32; ++ib) {
          if (32 * ib < n) {
            const int yoff = (32 * ib + 4 * L.mq + L.tq) * TB + 16 * cs0 + 4 * L.tp;
            const hx8 yh0 = dense_tr_pair(Tb + yoff, 16 * TB), yl0 = dense_tr_pair(Tb + yoff + TS, 16 * TB);
            const hx8 yh1 = dense_tr_pair(Tb + yoff + 16, 16 * TB), yl1 = dense_tr_pair(Tb + yoff + 16 + TS, 16 * TB);
            dense_kblock_mma2<NFT>(acc0, acc1, ihi, ilo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp + 16 * ft0, 16 * RS, yh0, yl0,
                                   yh1, yl1);
          }
        }
        dense_store_rows<NFT>(acc0, oscale, outb + 16 * ft0, (unsigned)hf, cs0 * 16 + L.mi, n, L);
        dense_store_rows<NFT>(acc1, oscale, outb + 16 * ft0, (unsigned)hf, cs0 * 16 + 16 + L.mi, n, L);
      }
    } else {
      if (wave < nstrips) {
        const int j = wave * 16 + L.mi;
        f32x4 acc[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ib = 0; ib < CW / 32; ++ib) {
          if (32 * ib < n) {
            const int yoff = (32 * ib + 4 * L.mq + L.tq) * TB + 16 * wave + 4 * L.tp;
            const hx8 yh = dense_tr_pair(Tb + yoff, 16 * TB);
            const hx8 yl = dense_tr_pair(Tb + yoff + TS, 16 * TB);
            dense_kblock_mma<F, 4>(acc, ihi, ilo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp, 16 * RS, yh, yl);
          }
        }
        if constexpr (FR == F) dense_store_rows<FT>(acc, oscale, outb, (unsigned)hf, j, n, L);
        else if (j < n) dense_store_acc<FT, true>(acc, oscale, outb, (unsigned)j * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
      }
    }
  };

  // ---- V:  dP^T = V dO^T -----------------------------------------------------------------------------------------------------
  image_commit(Kb);
  hx8 gh[KT], gl[KT];
  const float doinv = row_convert(gh, gl);
  row_fetch(Qb);
  lds_barrier();
  DFGNN_DSTAMP(1)
  f32x4 dS[U];
  {
    const LaneIds L = lane_ids();
    if (row_wave) {
      dense_rows_mma_strip<F, U>(dS, ihi, ilo, n, gh, gl, L);  // dP for now (x the two scales)
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) dS[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float dpc = isc.inv * doinv;
  DFGNN_DSTAMP(2)

  // ---- K:  S^T = K Q^T, P, t, dS, dQ^T = K^T dS^T;  P -> tile ----------------------------------------------------------------
  image_commit(dOb);
  hx8 qh[KT], ql[KT];
  const float qinv = row_convert(qh, ql);
  lds_barrier();
  DFGNN_DSTAMP(3)
  float tmax = 0.f;
  {
    const LaneIds L = lane_ids();
    f32x4 P[U];
    if (row_wave) {
      dense_rows_mma_strip<F, U>(P, ihi, ilo, n, qh, ql, L);  // S (x the two scales)
      const float c2 = (isc.inv * qinv) * kLog2e;
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned bits = (mwd[u / 2] >> (16 * (u & 1) + 4 * L.mq)) & 0xFu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = ((bits >> r) & 1u) ? __builtin_amdgcn_exp2f(fmaf(P[u][r], c2, -smx)) * sinv : 0.f;
          P[u][r] = p;
          dS[u][r] *= dpc;
          t = fmaf(p, dS[u][r], t);
        }
      }
      t = xor16_32_sum(t);  // a row lives on 4 lanes of this wave
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dS[u][r] = P[u][r] * (dS[u][r] - t);
          tmax = fmaxf(tmax, fabsf(dS[u][r]));
        }
      // dQ^T = K^T dS^T: the dS strip is the operand as it stands (under this strip's own scale)
      const Pow2Scale tw = pow2_scale(wave_max(tmax));
      f32x4 qacc[FT];
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) qacc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jb = 0; jb < CW / 32; ++jb)
        if (32 * jb < n) dense_cols_mma<F, 4>(qacc, ihi, ilo, jb, dS[2 * jb], dS[2 * jb + 1], tw.s, L);
      const int i = wave * 16 + L.mi;
      if constexpr (FR == F) dense_store_rows<FT>(qacc, isc.inv * tw.inv, dQb, (unsigned)hf, i, n, L);
      else if (i < n) dense_store_acc<FT, true>(qacc, isc.inv * tw.inv, dQb, (unsigned)i * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) P[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    strip_to_tile(P, kUnitScale);  // every wave its own 16 rows (zeros past the range): P lies in [0, 1]
  }
  wg_max_post(smax + kDenseWaves, tmax);  // the dS tile's scale needs the largest |dS| of the range
  DFGNN_DSTAMP(4)

  // ---- dO:  dV^T = dO^T P ------------------------------------------------------------------------------------------------------
  image_commit(Qb);
  lds_barrier();
  DFGNN_DSTAMP(5)
  column_product(dVb, isc.inv * kUnitScaleInv);
  DFGNN_DSTAMP(6)

  // ---- Q:  dS -> tile, dK^T = Q^T dS -----------------------------------------------------------------------------------------
  wg_max_post(smax, dense_stage_absmax<F, CW>(st));
  lds_barrier();  // the tile and the dO image are free
  const Pow2Scale ts = pow2_scale(wg_max_read(smax + kDenseWaves));
  strip_to_tile(dS, ts.s);
  isc = pow2_scale(wg_max_read(smax));
  dense_stage_store<F, CW>(st, ihi, ilo, isc.s, fr);
  lds_barrier();
  DFGNN_DSTAMP(7)
  column_product(dKb, isc.inv * ts.inv);
  DFGNN_DSTAMP(8)
}

}  // namespace dfgnn
