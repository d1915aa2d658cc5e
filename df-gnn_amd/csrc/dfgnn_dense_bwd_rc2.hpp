// dfgnn_dense_bwd_rc2.hpp -- matrix-core GT backward of the statistics-saving pair for dense ranges of <= 128 nodes:
// P recomputed, images staged in PAIRS, five image loads, no register operands from memory, no edge list.
//
// What bounds these kernels is the number of bytes a CU pulls through its memory pipe: the time of a
// range is nearly proportional to the bytes it loads, cached or not.  The attn_edge-based body loads 4 feature matrices +
// 6 bytes per edge; recomputing P costs a fifth matrix load at the very least, because S = Q K^T and dQ = dS K want K
// resident at the two ends of the dependency chain (S -> P -> dS) while dV = dO^T P and dK = Q^T dS want the matrices
// that S and dP take their ROWS from.  The first form of this body (one image at a time, the strips' dO and Q rows fetched
// a second time as 32-byte register pieces from memory: six loads, two of them half-line requests) measured 4 % slower
// than this one, and 21 - 36 us of its 145 - 165 us per 1024 ranges were those pieces (a build without them: DESIGN.md).
// Here the LDS holds TWO images (2 x 72 KB; the P / dS tile aliases the second):
//     [dO | V]   dP^T = V dO^T                  (the strips take their dO rows from the dO image)
//     [Q  | K]   S^T = K Q^T, P, t, dS           (... their Q rows from the Q image)
//                dQ^T = K^T dS^T                 (the dS accumulators ARE the operand: no tile)
//                tile := dS (over the K image),  dK^T = Q^T dS
//     [dO | P ]  tile := P,                      dV^T = dO^T P        (dO staged a second time: the fifth load)
// Two images travel at a time (twice the bytes in flight per CU), every load is a whole-line image load.
// P_ij = 2^(S_ij c - m_i c) / l_i on the edges of the plan's bitmap of row i (g.mask), (m_i, l_i) = the forward's row
// statistics.  Numerics and layouts: dfgnn_dense.hpp.
// Replaces, for such ranges, fused_gtconv_backward.cu:40-191 (with gt_dense_stats.hip's forward in place of :31-163).
#pragma once
#include "dfgnn_dense.hpp"
#include "dfgnn_dense_stamp.hpp"

namespace dfgnn {

// WEIGHTED (g.wdense: the plan's dense edge values): the logit of an edge is S val, and d logit / d S = val multiplies dS
// before it is used for dQ and dK (the reference's attn * val, fused_gtconv_hyper.cu:88-90, differentiated).
template <int FR, bool WEIGHTED = false>
__device__ __forceinline__ void dense_bwd_rc2_body(float *lds, const Csr &g, int n0, int n, int head,
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
  h16 *ahi = reinterpret_cast<h16 *>(lds), *alo = ahi + (size_t)CW * RS;  // image A: dO, Q, dO
  h16 *bhi = alo + (size_t)CW * RS, *blo = bhi + (size_t)CW * RS;        // image B: V, K -- or the tile
  constexpr size_t kImgB = 2 * (size_t)CW * RS, kTileH = (size_t)CW * TB;  // fp16 elements
  h16 *Tb = bhi;                                                          // the tile: CW rows of hi[TS] | lo[TS]
  float *smax = reinterpret_cast<float *>(bhi + (kImgB > kTileH ? kImgB : kTileH));  // [3][8]: image A, image B, dS
  DFGNN_LDS_AT(lds, (unsigned)(reinterpret_cast<char *>(smax + 3 * kDenseWaves) - reinterpret_cast<char *>(lds)));  // the carve-up fits
  const size_t hf = (size_t)g.h * fr, hoff = (size_t)head * fr;
  const float *Qb = Q + (size_t)n0 * hf + hoff, *Kb = K + (size_t)n0 * hf + hoff, *Vb = V + (size_t)n0 * hf + hoff,
              *dOb = dO + (size_t)n0 * hf + hoff;
  float *dQb = dQ + (size_t)n0 * hf + hoff, *dKb = dK + (size_t)n0 * hf + hoff, *dVb = dV + (size_t)n0 * hf + hoff;
  const bool row_wave = wave * 16 < n;

  DFGNN_DSTAMP(0)
  // ---- prologue: this lane's row bitmap and statistics (small, first), the dO and V images -----------------------------
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
  DenseStageRegs<F, CW> stA, stB;
  dense_stage_load<F, CW>(stA, dOb, hf, 0, n, fr);
  dense_stage_load<F, CW>(stB, Vb, hf, 0, n, fr);
  // this strip's rows of the image in A, as they stand there (fp16 hi / lo under the image's scale)
  auto strip_rows = [&](hx8 (&oh)[KT], hx8 (&ol)[KT]) {
    const LaneIds L = lane_ids();
    const int off = (wave * 16 + L.mi) * RS + 8 * L.mq;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      oh[t] = row_wave ? *reinterpret_cast<const hx8 *>(ahi + off + 32 * t) : hx8{};
      ol[t] = row_wave ? *reinterpret_cast<const hx8 *>(alo + off + 32 * t) : hx8{};
    }
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
  // out^T[f][c] = sum_i X[i][f] Y[i][c], X = the image in A, Y = the tile: wave w takes the column strips 2 (w / 2),
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
            dense_kblock_mma2<NFT>(acc0, acc1, ahi, alo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp + 16 * ft0, 16 * RS, yh0, yl0,
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
            dense_kblock_mma<F, 4>(acc, ahi, alo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp, 16 * RS, yh, yl);
          }
        }
        if constexpr (FR == F) dense_store_rows<FT>(acc, oscale, outb, (unsigned)hf, j, n, L);
        else if (j < n) dense_store_acc<FT, true>(acc, oscale, outb, (unsigned)j * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
      }
    }
  };

  // ---- [dO | V]:  dP^T = V dO^T ----------------------------------------------------------------------------------------------
  wg_max_post(smax, dense_stage_absmax<F, CW>(stA));
  wg_max_post(smax + kDenseWaves, dense_stage_absmax<F, CW>(stB));
  lds_barrier();
  Pow2Scale sa = pow2_scale(wg_max_read(smax)), sb = pow2_scale(wg_max_read(smax + kDenseWaves));
  dense_stage_store<F, CW>(stA, ahi, alo, sa.s, fr);
  dense_stage_store<F, CW>(stB, bhi, blo, sb.s, fr);
  dense_stage_load<F, CW>(stA, Qb, hf, 0, n, fr);  // the next pair travels during dP
  dense_stage_load<F, CW>(stB, Kb, hf, 0, n, fr);
  lds_barrier();
  DFGNN_DSTAMP(1)
  f32x4 dS[U];
  {
    hx8 gh[KT], gl[KT];
    strip_rows(gh, gl);
    const LaneIds L = lane_ids();
    if (row_wave) {
      dense_rows_mma_strip<F, U>(dS, bhi, blo, n, gh, gl, L);  // dP for now (x the two scales)
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) dS[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float dpc = sa.inv * sb.inv;
  DFGNN_DSTAMP(2)

  // ---- [Q | K]:  S^T = K Q^T, P, t, dS, dQ^T = K^T dS^T; then tile := dS, dK^T = Q^T dS ---------------------------------------
  wg_max_post(smax, dense_stage_absmax<F, CW>(stA));
  wg_max_post(smax + kDenseWaves, dense_stage_absmax<F, CW>(stB));
  lds_barrier();  // every wave is done with the dO and V images
  sa = pow2_scale(wg_max_read(smax));
  sb = pow2_scale(wg_max_read(smax + kDenseWaves));
  dense_stage_store<F, CW>(stA, ahi, alo, sa.s, fr);
  dense_stage_store<F, CW>(stB, bhi, blo, sb.s, fr);
  dense_stage_load<F, CW>(stA, dOb, hf, 0, n, fr);  // dO again, for dV: it travels during everything below
  lds_barrier();
  DFGNN_DSTAMP(3)
  float tmax = 0.f;
  f32x4 P[U];
  {
    hx8 qh[KT], ql[KT];
    strip_rows(qh, ql);
    const LaneIds L = lane_ids();
    if (row_wave) {
      dense_rows_mma_strip<F, U>(P, bhi, blo, n, qh, ql, L);  // S (x the two scales)
      const float c2 = (sa.inv * sb.inv) * kLog2e;
      // The exponentials are normalised by THEIR OWN row sum (the forward's l_i belongs to the forward's rounding of S:
      // with logits of +-100 the two differ by 1e-5 relative, and sum_j P_ij = 1 is what the softmax Jacobian assumes);
      // the forward's m_i only has to be near the maximum.  sinv just says whether the row has edges at all.
      float t = 0.f, l = 0.f;
      const float *wrow = WEIGHTED ? g.wdense + (size_t)(n0 + min(wave * 16 + L.mi, n - 1)) * kPlanWeightStride + 4 * L.mq : nullptr;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const unsigned bits = (mwd[u / 2] >> (16 * (u & 1) + 4 * L.mq)) & 0xFu;
        if constexpr (WEIGHTED) {
          const float4 wv = (16 * u < n) ? ld32_f4(wrow, 16u * u) : make_float4(0.f, 0.f, 0.f, 0.f);
          P[u][0] *= wv.x; P[u][1] *= wv.y; P[u][2] *= wv.z; P[u][3] *= wv.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = ((bits >> r) & 1u) ? __builtin_amdgcn_exp2f(fmaf(P[u][r], c2, -smx)) : 0.f;
          P[u][r] = p;
          l += p;
          dS[u][r] *= dpc;
          t = fmaf(p, dS[u][r], t);
        }
      }
      l = xor16_32_sum(l);  // a row lives on 4 lanes of this wave
      const float linv = (sinv != 0.f && l > 0.f) ? 1.f / l : 0.f;
      t = xor16_32_sum(t) * linv;
#pragma unroll
      for (int u = 0; u < U; ++u) P[u] *= linv;
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) dS[u][r] = P[u][r] * (dS[u][r] - t);
      if constexpr (WEIGHTED) {  // (dS is zero off the edges: whatever the weight rows hold there is not read as a factor of a non-zero)
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const unsigned bits = (mwd[u / 2] >> (16 * (u & 1) + 4 * L.mq)) & 0xFu;
          const float4 wv = (16 * u < n) ? ld32_f4(wrow, 16u * u) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float wvr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) dS[u][r] = ((bits >> r) & 1u) ? dS[u][r] * wvr[r] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, fabsf(dS[u][r]));
      // dQ^T = K^T dS^T: the dS strip is the operand as it stands (under this strip's own scale)
      const Pow2Scale tw = pow2_scale(wave_max(tmax));
      f32x4 qacc[FT];
#pragma unroll
      for (int ft = 0; ft < FT; ++ft) qacc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jb = 0; jb < CW / 32; ++jb)
        if (32 * jb < n) dense_cols_mma<F, 4>(qacc, bhi, blo, jb, dS[2 * jb], dS[2 * jb + 1], tw.s, L);
      const int i = wave * 16 + L.mi;
      if constexpr (FR == F) dense_store_rows<FT>(qacc, sb.inv * tw.inv, dQb, (unsigned)hf, i, n, L);
      else if (i < n) dense_store_acc<FT, true>(qacc, sb.inv * tw.inv, dQb, (unsigned)i * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) P[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  wg_max_post(smax + 2 * kDenseWaves, tmax);  // the dS tile's scale needs the largest |dS| of the range
  lds_barrier();  // every wave is done with the K image: the tile takes its place
  DFGNN_DSTAMP(4)
  const Pow2Scale ts = pow2_scale(wg_max_read(smax + 2 * kDenseWaves));
  strip_to_tile(dS, ts.s);
  lds_barrier();
  column_product(dKb, sa.inv * ts.inv);
  DFGNN_DSTAMP(5)

  // ---- [dO | P]:  tile := P, dV^T = dO^T P --------------------------------------------------------------------------------------
  wg_max_post(smax, dense_stage_absmax<F, CW>(stA));
  lds_barrier();  // the Q image and the dS tile are free
  sa = pow2_scale(wg_max_read(smax));
  dense_stage_store<F, CW>(stA, ahi, alo, sa.s, fr);
  strip_to_tile(P, kUnitScale);  // every wave its own 16 rows (zeros past the range): P lies in [0, 1]
  lds_barrier();
  DFGNN_DSTAMP(6)
  column_product(dVb, sa.inv * kUnitScaleInv);
  DFGNN_DSTAMP(7)
}

}  // namespace dfgnn
