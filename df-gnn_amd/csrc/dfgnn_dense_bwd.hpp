// dfgnn_dense_bwd.hpp -- the backward body of the matrix-core GT / GAT kernels (gt_dense.hip, gt_dense_stats.hip).
// See gt_dense.hip for the design, dfgnn_dense.hpp for the numerics and the operand layouts.
#pragma once
#include "dfgnn_dense_fwd.hpp"

namespace dfgnn {

// =====================================================================================================================
// backward
// =====================================================================================================================
// NBLK column blocks of CW columns each; rows are processed in blocks of RB rows (dP / P / dS of every column block
// of a row block stay in registers):
//   (CW, NBLK) = (128, 1)  up to 128 nodes : one 128 x 128 tile
//              = (160, 1)  up to 160 nodes : two row blocks of <= 80 rows against all (<= 160) columns
//              = (128, 2)  up to 255 nodes : two row blocks of 128 rows x two column blocks
template <int CW, int NBLK>
struct DenseBwdGeom {
  static constexpr int RB = (CW == kDenseWideRows) ? 80 : 128;  // rows per row block
  static constexpr int RBP = (RB + 31) & ~31;                  // ... padded to the 32-deep k-blocks of the products
  static constexpr int U = CW / 16;                            // 16-column tiles per column block
  static constexpr int TS = CW + 8;                            // floats per tile row == 2 x TS fp16 (hi | lo)
};

// GAT training backward (GAT = true; Q = attn_row [m, h], K = attn_col [m, h], V = X, dV = grad_feat; attn_edge, dQ, dK
// unused): P is recomputed per edge from the row statistics of the forward (staged in LDS next to the tile), the two
// feature products are the GT ones (grad_feat^T = dO^T P, dP^T = X dO^T), and instead of the dQ / dK products
// G = dS LeakyReLU'(attn_row[i] + attn_col[j]) is summed over rows (registers) and columns (per-strip partial sums
// through the tile, which is free by then; fixed summation order, no atomics).
struct GatBwdArgs {
  const float *edge_max, *edge_sum;  // [m, h] from the training forward
  float slope;
  float *grad_row, *grad_col;        // [m, h]
  DenseDrop drop;                    // attention dropout: a dropped edge enters the P tile with a negative sign
};

// RECOMP (GT backward of the statistics-saving pair, gt_dense_stats.hip): there is no attn_edge.  P is recomputed --
// S^T = K Q^T on the matrix cores for the strip's rows (one more K image per column block, ahead of the dO image; the
// strip's Q rows come straight from memory as register operands, as in the forward), masked with the plan's edge bitmap
// of the lane's row (g.mask), p = 2^(S c - max c) / sum with the forward's row statistics (ga.edge_max / ga.edge_sum) --
// and written to the tile by its strips; no edge list, no scatter, no clearing of the tile.
// WEIGHTED (with RECOMP; g.wdense = the plan's dense edge values): logits S val, and dS val for dQ / dK (see dense_bwd_rc2_body).
template <int FR, int CW, int NBLK, bool GAT = false, bool RECOMP = false, bool WEIGHTED = false>
__device__ __forceinline__ void dense_bwd_body(float *lds, const Csr &g, int n0, int n, int e0, int ne, int head,
                                               const float *__restrict__ Q, const float *__restrict__ K,
                                               const float *__restrict__ V, const float *__restrict__ attn_edge,
                                               const float *__restrict__ dO, float *__restrict__ dQ,
                                               float *__restrict__ dK, float *__restrict__ dV,
                                               const GatBwdArgs ga = GatBwdArgs{}) {
  static_assert(!(RECOMP && (GAT || CW != kDenseChunkRows)), "P is recomputed for GT, 128-column blocks");
  constexpr int F = FR < 32 ? 32 : FR;  // layout width (see dense_fwd_body)
  constexpr int fr = FR;
  using D = DenseCfg<F>;
  using G = DenseBwdGeom<CW, NBLK>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT, RB = G::RB, RBP = G::RBP, U = G::U, TS = G::TS;
  constexpr int TB = 2 * TS;  // fp16 elements per interleaved tile row: hi at +0, lo at +TS
  // edges fetched ahead per thread (GAT: each edge also carries its dropout random, so fewer fit the registers)
  constexpr int PRE = GAT ? 10 : kDensePre;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);  // = this wave's strip of a row block
  h16 *ihi = reinterpret_cast<h16 *>(lds), *ilo = ihi + (size_t)CW * RS;
  float *T = reinterpret_cast<float *>(ilo + (size_t)CW * RS);
  h16 *Tb = reinterpret_cast<h16 *>(T);
  const size_t hf = (size_t)g.h * fr, hoff = (size_t)head * fr;
  const float *Qb = Q + (size_t)n0 * hf + hoff, *Kb = K + (size_t)n0 * hf + hoff, *Vb = V + (size_t)n0 * hf + hoff,
              *dOb = dO + (size_t)n0 * hf + hoff;
  float *dQb = dQ + (size_t)n0 * hf + hoff, *dKb = dK + (size_t)n0 * hf + hoff, *dVb = dV + (size_t)n0 * hf + hoff;
  const float *attn_h = RECOMP ? nullptr : attn_edge + (size_t)head * g.nnz;
  // GAT: per-node scalars of the range, [SN] each, behind the tile: attn_row, attn_col, edge_max, 1 / edge_sum
  constexpr int SN = NBLK * CW;
  float *smax = T + RBP * TS;            // [8] per-wave maxima of the image being staged, [8] of the dS tile
  float *arl = smax + 2 * kDenseWaves, *acl = arl + SN, *mxl = acl + SN, *ivl = mxl + SN;
  (void)arl, (void)acl, (void)mxl, (void)ivl;
  DFGNN_LDS_AT(lds, (unsigned)(reinterpret_cast<char *>(GAT ? ivl + SN : arl) - reinterpret_cast<char *>(lds)));  // the carve-up fits

  DFGNN_DSTAMP(0)
  if constexpr (GAT) {
    const int tid = opaque_tid();
    if (tid < SN) {
      const size_t k = (size_t)(n0 + min(tid, n - 1)) * g.h + head;
      const float a = Q[k], c = K[k], mx = ga.edge_max[k], sm = ga.edge_sum[k];
      const bool valid = tid < n;
      arl[tid] = valid ? a : 0.f;
      acl[tid] = valid ? c : 0.f;
      mxl[tid] = valid ? mx : 0.f;
      ivl[tid] = (valid && sm != 0.f) ? 1.f / sm : 0.f;
    }  // (visible to the first scatter: load_tile has a barrier between zeroing the tile and scattering)
  }
  // The image that is needed next is fetched one phase ahead into registers (`st`) -- for a single column block; with
  // two column blocks the registers hold dP / P / dS of both and the image is fetched where it is stored.
  DenseStageRegs<F, CW> st;
  const float *next_src = nullptr;
  int next_row0 = 0, next_end = 0;
  auto image_prefetch = [&](const float *src, int row0, int row_end, bool once = false) {  // once: see dense_stage_load
    if (NBLK > 1) {
      next_src = src;
      next_row0 = row0;
      next_end = row_end;
    } else {
      dense_stage_load<F, CW>(st, src, hf, row0, row_end, fr, once);
    }
  };
  // An image goes to LDS in two steps around a barrier the phase structure has anyway: image_post() (with two column
  // blocks: fetch it now) posts this wave's largest magnitude, image_store() -- after the barrier -- derives the
  // image's power-of-two scale from all of them and stores the fp16 halves.  `isc` = scale of the resident image.
  Pow2Scale isc{1.f, 1.f};
  auto image_post = [&]() {
    if (NBLK > 1) dense_stage_load<F, CW>(st, next_src, hf, next_row0, next_end, fr);
    wg_max_post(smax, dense_stage_absmax<F, CW>(st));
  };
  auto image_store = [&]() {
    isc = pow2_scale(wg_max_read(smax));
    dense_stage_store<F, CW>(st, ihi, ilo, isc.s, fr);
  };
  // Edges of a row block (CSR order, contiguous) are fetched several per thread at a time -- loads first, then the
  // scatter into the fp32 tile -- so that a batch costs one memory round trip; the first batch of a range is
  // fetched ahead of everything else.
  unsigned pc[PRE];  // packed (row, column) within the range (plan.hip: coords)
  float pa[PRE];
  auto edges_prefetch = [&](int ea, int eb) {  // first PRE edges per thread of the row block [ea, eb)
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      // clamped: plain loads (a row block without edges -- rows with in-edges only -- reads the last edge, unused)
      const unsigned e = (unsigned)min(tid + k * kDenseThreads, max(eb - ea, 1) - 1);
      const int ea0 = min(ea, g.nnz - 1);
      pc[k] = ld32_once(g.coords + ea0, e);
      if constexpr (!GAT) pa[k] = ld32_once(attn_h + ea0, e);
      else pa[k] = ga.drop.mask ? ga.drop.mask[((size_t)ea0 + e) * g.h + head] : 1.f;
    }
  };
  // GAT: P of edge (row i, column j of the range) from the staged scalars
  auto gat_p = [&](int i, int j, float rnd) {
    const float pp = fast_exp(leaky_relu(arl[i] + acl[j], ga.slope) - mxl[i]) * ivl[i];
    return (rnd > ga.drop.drop) ? pp : -pp;  // (no dropout: rnd = 1, drop = 0)
  };
  // The P tile of (row block i0, column block j0), fp32: zero it, scatter the edges [ea, eb) of the row block into it
  // and, if `commit`, put the prefetched image into LDS.  The first PRE edges per thread of a range's first tile were
  // fetched in the prologue (pi, pj, pa) and are scattered ahead of the row-block loop (tile_open): used inside it they
  // would be live -- and spilled -- around the whole loop.
  // One tile, GT: P goes into the tile as fp16 hi | lo halves (scale 2^14) straight from the scatter -- the form the column
  // product wants -- and the strips read their rows back from it (hi + lo = P to 2^-24); no fp32 copy, no conversion
  // pass, one barrier less.  (Two column blocks / GAT: P is scattered as fp32 and converted in place by its strips.)
  constexpr bool kDirectP = !GAT && NBLK == 1 && !RECOMP;
  auto put_p = [&](int i, int j, float p) {
    const h16 hh = (h16)(p * kUnitScale);
    Tb[i * TB + j] = hh;
    Tb[i * TB + TS + j] = (h16)fmaf(p, kUnitScale, -(float)hh);
  };
  auto tile_clear = [&]() {
    const int tid = opaque_tid();
    for (int k = tid; k < RBP * TS / 4; k += kDenseThreads)
      reinterpret_cast<float4 *>(T)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    lds_barrier();
  };
  auto tile_open = [&](int ea, int eb) {  // first tile of the range: (i0, j0) = (0, 0)
    tile_clear();
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int i = pc[k] >> 8, j = pc[k] & 0xFF;
      if (tid + k * kDenseThreads < eb - ea && j < CW) {
        if constexpr (GAT) T[i * TS + j] = gat_p(i, j, pa[k]);
        else if constexpr (kDirectP) put_p(i, j, pa[k]);
        else T[i * TS + j] = pa[k];
      }
    }
  };
  auto load_tile = [&](int i0, int j0, int ea, int eb, bool opened, bool commit) {
    const int tid = opaque_tid();
    if (!opened) tile_clear();
    constexpr int B = 8;
    for (int base = opened ? PRE * kDenseThreads : 0; base < eb - ea; base += B * kDenseThreads) {
      unsigned bc[B];
      float ba[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const unsigned e = (unsigned)min(base + tid + k * kDenseThreads, eb - ea - 1);  // (eb > ea inside this loop)
        bc[k] = ld32_once(g.coords + ea, e);
        if constexpr (!GAT) ba[k] = ld32_once(attn_h + ea, e);
        else ba[k] = ga.drop.mask ? ga.drop.mask[((size_t)ea + e) * g.h + head] : 1.f;
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const int i = bc[k] >> 8, jj = bc[k] & 0xFF, j = jj - j0;
        if (base + tid + k * kDenseThreads < eb - ea && j >= 0 && j < CW) {
          if constexpr (GAT) T[(i - i0) * TS + j] = gat_p(i, jj, ba[k]);
          else if constexpr (kDirectP) put_p(i - i0, j, ba[k]);
          else T[(i - i0) * TS + j] = ba[k];
        }
      }
    }
    if (commit) {  // after the scatter: the edge loads were issued before the image's
      image_post();
      lds_barrier();
      image_store();
    }
    lds_barrier();
  };
  // this strip's 16 x CW values (times the tile's power-of-two scale) -> its own rows of the tile, as interleaved fp16
  // hi | lo halves
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
  // out^T[f][c] = sum_i X[i][f] Y[i][c]: X = the image (ni rows), Y = the fp16 tile, c = the 16 columns of column
  // strip cs (rows j0 + 16 cs .. of the output); oscale = 1 / (image scale x tile scale)
  auto column_strip = [&](float *outb, int j0, int cs, int ni, bool accumulate, float oscale) {
    const LaneIds L = lane_ids();
    const int j = j0 + cs * 16 + L.mi;
    f32x4 acc[FT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ib = 0; ib < RBP / 32; ++ib) {
      if (32 * ib < ni) {
        const int yoff = (32 * ib + 4 * L.mq + L.tq) * TB + 16 * cs + 4 * L.tp;
        const hx8 yh = dense_tr_pair(Tb + yoff, 16 * TB);
        const hx8 yl = dense_tr_pair(Tb + yoff + TS, 16 * TB);
        dense_kblock_mma<F, (NBLK == 1 ? 8 : 4)>(acc, ihi, ilo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp, 16 * RS, yh, yl);
      }
    }
    if constexpr (FR == F) {
      if (!accumulate) {  // (wave-uniform) whole-line stores
        dense_store_rows<FT>(acc, oscale, outb, (unsigned)hf, j, n, L);
        return;
      }
    }
    if (j < n) dense_store_acc<FT, (FR < F)>(acc, oscale, outb, (unsigned)j * (unsigned)hf + 4u * L.mq, accumulate, 4 * L.mq, fr);
  };
  // one 16 x 16 output tile (column strip cs, feature tile ft): the unit of work for the strips past the eighth,
  // which are dealt out tile by tile so that all waves share them
  auto column_tile = [&](float *outb, int j0, int cs, int ft, int ni, bool accumulate, float oscale) {
    const LaneIds L = lane_ids();
    const int j = j0 + cs * 16 + L.mi;
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ib = 0; ib < RBP / 32; ++ib) {
      if (32 * ib < ni) {
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
      dense_store_acc<1, (FR < F)>(acc, oscale, outb, (unsigned)j * (unsigned)hf + 16u * ft + 4u * L.mq, accumulate, 16 * ft + 4 * L.mq, fr);
  };
  // the same product for a single tile of <= 128 x 128: wave w takes the column strips 2 (w / 2), 2 (w / 2) + 1 and the
  // feature tiles of half w % 2 (dense_kblock_mma2: the image fragments are shared by the two strips)
  constexpr bool kBlocked = NBLK == 1 && U == kDenseWaves && FR == F && FT >= 4;
  auto column_block = [&](float *outb, int ni, float oscale) {
    constexpr int NFT = kBlocked ? FT / 2 : 2;  // (compiled for every instance, used by the blocked ones)
    const LaneIds L = lane_ids();
    const int cs0 = 2 * (wave >> 1), ft0 = NFT * (wave & 1);
    f32x4 acc0[NFT], acc1[NFT];
#pragma unroll
    for (int k = 0; k < NFT; ++k) acc0[k] = acc1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ib = 0; ib < RBP / 32; ++ib) {
      if (32 * ib < ni) {
        const int yoff = (32 * ib + 4 * L.mq + L.tq) * TB + 16 * cs0 + 4 * L.tp;
        const hx8 yh0 = dense_tr_pair(Tb + yoff, 16 * TB), yl0 = dense_tr_pair(Tb + yoff + TS, 16 * TB);
        const hx8 yh1 = dense_tr_pair(Tb + yoff + 16, 16 * TB), yl1 = dense_tr_pair(Tb + yoff + 16 + TS, 16 * TB);
        dense_kblock_mma2<NFT>(acc0, acc1, ihi, ilo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp + 16 * ft0, 16 * RS, yh0,
                               yl0, yh1, yl1);
      }
    }
    dense_store_rows<NFT>(acc0, oscale, outb + 16 * ft0, (unsigned)hf, cs0 * 16 + L.mi, n, L);
    dense_store_rows<NFT>(acc1, oscale, outb + 16 * ft0, (unsigned)hf, cs0 * 16 + 16 + L.mi, n, L);
  };
  auto column_phase = [&](float *outb, int j0, int ni, bool accumulate, float oscale) {
    const int nstrips = min(U, (n - j0 + 15) >> 4);
    if constexpr (kBlocked) {  // (one tile: j0 = 0, nothing to accumulate onto)
      if (2 * (wave >> 1) < nstrips) column_block(outb, ni, oscale);
      return;
    }
    if (wave < nstrips) column_strip(outb, j0, wave, ni, accumulate, oscale);
    if (U > kDenseWaves)
      for (int unit = wave; unit < (nstrips - kDenseWaves) * FT; unit += kDenseWaves)
        column_tile(outb, j0, kDenseWaves + unit / FT, unit % FT, ni, accumulate, oscale);
  };

  int ea = e0, eb = (!RECOMP && RB < n) ? g.row_ptr[n0 + RB] : e0 + ne;  // edges of the current row block
  if constexpr (RECOMP) {
    image_prefetch(Kb, 0, n);  // the first image: K rows of column block 0
  } else {
    edges_prefetch(ea, eb);
    image_prefetch(dOb, 0, min(n, RB), !RECOMP && !GAT);  // (the attn_edge form reads dO once)
    tile_open(ea, eb);
  }
  float gcol = 0.f;  // GAT: grad_attn_col of column opaque_tid(), accumulated over the row blocks
  constexpr int NRB = (NBLK * CW + RB - 1) / RB;  // row blocks at most (one for a single tile: then this is no loop)
  for (int rb = 0; rb < NRB && rb * RB < n; ++rb) {
    const int i0 = rb * RB;
    const int ni = min(n - i0, RB);
    const bool row_wave = wave * 16 < ni;
    const bool first = i0 == 0;

    // ---- dV^T = dO^T P, column block by column block; the strips pick up their dO rows (the register operand of
    //      dP) and their P values on the way: dO is read from global memory once ---------------------------------------
    f32x4 dS[NBLK][U], Pr[NBLK][U];
    hx8 gh[KT], gl[KT];
    float doinv = 1.f;  // 1 / scale of this row block's dO image (and of gh / gl, which are read from it)
    if constexpr (RECOMP) {
      // ---- P of the row block, recomputed: S^T = K Q^T against every column block ------------------------------------
      const LaneIds L = lane_ids();
      const int irow = i0 + wave * 16 + L.mi;  // this lane's row of the range
      float4 qa[KT], qb[KT];                   // its pieces of that row of Q, raw
      {
        const unsigned off = (unsigned)min(irow, n - 1) * (unsigned)hf;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          const unsigned c = (FR == F || 32 * t + 8 * L.mq < fr) ? 32u * t + 8u * L.mq : 0u;  // (past fr: zeroed below)
          qa[t] = ld32_f4(Qb, off + c);
          qb[t] = ld32_f4(Qb, off + c + 4);
        }
      }
      constexpr int MW = (NBLK * U + 1) / 2;  // bitmap words of a row
      unsigned mwd[MW];
      float smx, sinv;
      {
        const size_t node = (size_t)(n0 + min(irow, n - 1));
        const unsigned *mp = g.mask + node * kPlanMaskWords;
#pragma unroll
        for (int w = 0; w < MW; ++w) mwd[w] = (row_wave && irow < n) ? ld32(mp, (unsigned)w) : 0u;
        smx = ga.edge_max[node * g.h + head];
        const float ssum = ga.edge_sum[node * g.h + head];
        sinv = (ssum != 0.f) ? 1.f / ssum : 0.f;
      }
      hx8 qh[KT], ql[KT];
      float qinv = 1.f, lsum = 0.f;  // lsum: this lane's part of the row's sum of exponentials
#pragma unroll
      for (int jc = 0; jc < NBLK; ++jc) {
        if (jc > 0) image_prefetch(Kb, jc * CW, n);
        image_post();
        lds_barrier();  // the previous image is free
        image_store();  // K rows of column block jc
        if (jc + 1 == NBLK) image_prefetch(dOb, i0, i0 + ni);  // next image: the dO rows of this row block
        if (jc == 0) {  // the strip's Q rows as fp16 halves under their own power-of-two scale
          const bool valid = row_wave && irow < n;
          float qm = 0.f;
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            if (!valid || (FR < F && 32 * t + 8 * L.mq >= fr)) qa[t] = qb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            qm = fmaxf(qm, absmax8(qa[t], qb[t]));
          }
          const Pow2Scale qs = pow2_scale(wave_max(qm));
          qinv = qs.inv;
#pragma unroll
          for (int t = 0; t < KT; ++t) split_hx8(qa[t], qb[t], qs.s, qh[t], ql[t]);
        }
        lds_barrier();
        if (row_wave) {
          const int nj = n - jc * CW;
          f32x4 S[U];
          if constexpr (NBLK == 1) {
            dense_rows_mma_strip<F, U>(S, ihi, ilo, nj, qh, ql, L);
          } else {
#pragma unroll
            for (int u = 0; u < U; ++u) S[u] = (16 * u < nj) ? dense_rows_mma<F>(ihi, ilo, u, qh, ql, L) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
          // p = 2^(S c - max c) / sum on the edges (the forward's formula), 0 elsewhere
          const float c2 = (isc.inv * qinv) * 1.4426950408889634f, b2 = smx * 1.4426950408889634f;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int ug = jc * U + u;
            const unsigned bits = (mwd[ug / 2] >> (16 * (ug & 1) + 4 * L.mq)) & 0xFu;
            if constexpr (WEIGHTED) {
              const float4 wv = ld32_f4(g.wdense + (size_t)(n0 + min(irow, n - 1)) * kPlanWeightStride + 4 * L.mq, 16u * ug);
              S[u][0] *= wv.x; S[u][1] *= wv.y; S[u][2] *= wv.z; S[u][3] *= wv.w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Pr[jc][u][r] = ((bits >> r) & 1u) ? __builtin_amdgcn_exp2f(fmaf(S[u][r], c2, -b2)) : 0.f;
              lsum += Pr[jc][u][r];
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) Pr[jc][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      // normalised by their own row sum (see dense_bwd_rc2_body): sum_j P_ij = 1 for the P that is differentiated
      lsum = xor16_32_sum(lsum);
      const float linv = (sinv != 0.f && lsum > 0.f) ? 1.f / lsum : 0.f;
#pragma unroll
      for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
        for (int u = 0; u < U; ++u) Pr[jc][u] *= linv;
    }
#pragma unroll
    for (int jc = 0; jc < NBLK; ++jc) {
      const int j0 = jc * CW;
      if constexpr (RECOMP) {
        strip_to_tile(Pr[jc], kUnitScale);  // every wave its own 16 rows (zeros past the row block): P lies in [0, 1]
        if (jc == 0) {
          image_post();   // the dO rows of this row block
          lds_barrier();  // the K image is free
          image_store();
          doinv = isc.inv;
        }
        if (jc + 1 == NBLK) image_prefetch(Vb, 0, n);  // next image: V rows 0..
        lds_barrier();
        if (jc == 0) {
          const LaneIds L = lane_ids();
          const int off = (wave * 16 + L.mi) * RS + 8 * L.mq;
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            gh[t] = row_wave ? *reinterpret_cast<const hx8 *>(ihi + off + 32 * t) : hx8{};
            gl[t] = row_wave ? *reinterpret_cast<const hx8 *>(ilo + off + 32 * t) : hx8{};
          }
        }
      } else {
      load_tile(i0, j0, ea, eb, first && jc == 0, jc == 0);  // tile = P (fp32); image = dO rows of this row block
      if (jc == 0) doinv = isc.inv;
      DFGNN_DSTAMP(9)
      if (jc + 1 == NBLK) image_prefetch(Vb, 0, n);  // next image: V rows 0..
      if (row_wave) {
        const LaneIds L = lane_ids();
        if (jc == 0) {
          const int off = (wave * 16 + L.mi) * RS + 8 * L.mq;
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            gh[t] = *reinterpret_cast<const hx8 *>(ihi + off + 32 * t);
            gl[t] = *reinterpret_cast<const hx8 *>(ilo + off + 32 * t);
          }
        }
        const int nj = n - j0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (16 * u < nj) {
            if constexpr (kDirectP) {
              const h16 *trow = Tb + (wave * 16 + L.mi) * TB + 16 * u + 4 * L.mq;
              const hx4 h4 = *reinterpret_cast<const hx4 *>(trow), l4 = *reinterpret_cast<const hx4 *>(trow + TS);
#pragma unroll
              for (int r = 0; r < 4; ++r) Pr[jc][u][r] = ((float)h4[r] + (float)l4[r]) * kUnitScaleInv;
            } else {
              const float4 p = *reinterpret_cast<const float4 *>(T + (wave * 16 + L.mi) * TS + 16 * u + 4 * L.mq);
              Pr[jc][u] = f32x4{p.x, p.y, p.z, p.w};
            }
          } else {
            Pr[jc][u] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        if constexpr (GAT) {  // the tile of the grad_feat product holds the dropped-out attention
          f32x4 Pd[U];
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) Pd[u][r] = Pr[jc][u][r] > 0.f ? Pr[jc][u][r] : 0.f;  // (x drop.scale at the store)
          strip_to_tile(Pd, kUnitScale);
        } else if constexpr (!kDirectP) {
          strip_to_tile(Pr[jc], kUnitScale);  // in place, own rows only; P lies in [0, 1]
        }
      } else {  // (defined on every path: otherwise the arrays are carried around the row-block loop in registers)
#pragma unroll
        for (int u = 0; u < U; ++u) Pr[jc][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (jc == 0) {
#pragma unroll
          for (int t = 0; t < KT; ++t) gh[t] = gl[t] = hx8{};
        }
      }
      }  // (!RECOMP)
      if constexpr (!kDirectP && !RECOMP) lds_barrier();  // (the strips' in-place conversions)
      DFGNN_DSTAMP(3)
      column_phase(dVb, j0, ni, !first, doinv * kUnitScaleInv * (GAT ? ga.drop.scale : 1.f));
      if (jc + 1 == NBLK) image_post();  // V rows 0..
      lds_barrier();  // tile free (and, after the last block, the dO image)
    }
    DFGNN_DSTAMP(4)

    // ---- dP^T = V dO^T for every column block, t, dS ---------------------------------------------------------------
    float dpinv[NBLK];  // dP = acc x 1 / (V image scale x dO scale)
#pragma unroll
    for (int jc = 0; jc < NBLK; ++jc) {
      image_store();  // V rows of column block jc
      dpinv[jc] = isc.inv * doinv;
      if (jc + 1 < NBLK) {
        image_prefetch(Vb, (jc + 1) * CW, n);
      } else if constexpr (!GAT) {
        image_prefetch(Kb, 0, n);  // next image: K rows 0..
      } else if constexpr (CW * NBLK > RB) {
        // GAT: the next row block's dO rows are all that is left to fetch (its edges are fetched after dS: they would
        // not fit next to dP / P).  Unconditional -- after the last row block it re-reads one row (unused): a prefetch
        // under a condition turns the staging registers into loop-carried values and spills them.
        const bool more = rb + 1 < NRB && i0 + RB < n;  // (last block: every load is clamped onto row i0 -- cache hits, no HBM traffic)
        image_prefetch(dOb, more ? i0 + RB : i0, more ? min(n, i0 + 2 * RB) : i0 + 1);
      }
      lds_barrier();
      if (row_wave) {
        const LaneIds L = lane_ids();
        const int nj = n - jc * CW;
        if constexpr (NBLK == 1) {
          dense_rows_mma_strip<F, U>(dS[jc], ihi, ilo, nj, gh, gl, L);  // dP for now (double-buffered fragments)
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u)
            dS[jc][u] = (16 * u < nj) ? dense_rows_mma<F>(ihi, ilo, u, gh, gl, L) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) dS[jc][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (jc + 1 < NBLK) {
        image_post();   // V rows of the next column block
        lds_barrier();  // the next image overwrites this one
      }
    }
    DFGNN_DSTAMP(1)
    float tmax = 0.f;  // largest |dS| of this strip
    if (row_wave) {
      if constexpr (GAT) {  // g = keep dP / (1 - drop); P = |tile value|
#pragma unroll
        for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
          for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dS[jc][u][r] = Pr[jc][u][r] > 0.f ? dS[jc][u][r] * (dpinv[jc] * ga.drop.scale) : 0.f;
              Pr[jc][u][r] = fabsf(Pr[jc][u][r]);
            }
      } else {
#pragma unroll
        for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
          for (int u = 0; u < U; ++u) dS[jc][u] *= dpinv[jc];
      }
      float t = 0.f;
#pragma unroll
      for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) t = fmaf(Pr[jc][u][r], dS[jc][u][r], t);
      t = xor16_32_sum(t);  // a row lives on 4 lanes of this wave
#pragma unroll
      for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) dS[jc][u][r] = Pr[jc][u][r] * (dS[jc][u][r] - t);
      if constexpr (WEIGHTED) {  // d logit / d S = val (P is exactly zero off the edges)
        const LaneIds L = lane_ids();
        const float *wrow = g.wdense + (size_t)(n0 + min(i0 + wave * 16 + L.mi, n - 1)) * kPlanWeightStride + 4 * L.mq;
#pragma unroll
        for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float4 wv = ld32_f4(wrow, 16u * (jc * U + u));
            const float wvr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) dS[jc][u][r] = (Pr[jc][u][r] != 0.f) ? dS[jc][u][r] * wvr[r] : 0.f;
          }
      }
#pragma unroll
      for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, fabsf(dS[jc][u][r]));
    }
    if constexpr (!GAT) {
      wg_max_post(smax + kDenseWaves, tmax);  // the dS tile's scale needs the largest |dS| of the row block
      image_post();                           // K rows 0..
    }
    lds_barrier();  // the V image is free (and the two maxima are posted)
    DFGNN_DSTAMP(2)

    if constexpr (GAT) {
      // ---- G = dS LeakyReLU'(pre): row sums -> grad_attn_row, per-strip column sums -> the tile -> grad_attn_col -------
      float *cpart = T;  // [kDenseWaves][SN]
      if (rb + 1 < NRB && i0 + RB < n) {
        ea = eb;
        eb = (i0 + 2 * RB < n) ? g.row_ptr[n0 + i0 + 2 * RB] : e0 + ne;
      }
      if (row_wave) {
        const LaneIds L = lane_ids();
        const int i = i0 + wave * 16 + L.mi;
        const float ari = arl[min(i, n - 1)];
        float rs = 0.f;
#pragma unroll
        for (int jc = 0; jc < NBLK; ++jc)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const float4 a = *reinterpret_cast<const float4 *>(acl + jc * CW + 16 * u + 4 * L.mq);
            const float av[4] = {a.x, a.y, a.z, a.w};
            float cs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float ge = dS[jc][u][r] * ((ari + av[r] > 0.f) ? 1.f : ga.slope);
              rs += ge;
              cs[r] = lanes_sum<16>(ge);  // over the strip's 16 rows (the 16 lanes of a DPP row share mq)
            }
            if (L.mi == 0)
              *reinterpret_cast<float4 *>(cpart + wave * SN + jc * CW + 16 * u + 4 * L.mq) = make_float4(cs[0], cs[1], cs[2], cs[3]);
          }
        rs = xor16_32_sum(rs);
        if (L.mq == 0 && i < i0 + ni) ga.grad_row[(size_t)(n0 + i) * g.h + head] = rs;
      }
      lds_barrier();
      {
        const int tid = opaque_tid();
        if (tid < SN)
          for (int w = 0; w * 16 < ni; ++w) gcol += cpart[w * SN + tid];
      }
      lds_barrier();  // the next row block zeroes the tile
      continue;
    }

    // ---- dQ^T = K^T dS^T (accumulated over the column blocks in registers) and dK^T = Q^T dS ---------------------------
    f32x4 qacc[FT];  // in units of 1 / (dS tile scale x current K image scale)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) qacc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    const Pow2Scale ts = pow2_scale(wg_max_read(smax + kDenseWaves));  // scale of the dS tile(s) of this row block
    float kinv = 1.f;  // 1 / scale of the K block qacc is accumulated under
#pragma unroll
    for (int jc = 0; jc < NBLK; ++jc) {
      const int j0 = jc * CW, nj = min(n - j0, CW);
      if (row_wave) strip_to_tile(dS[jc], ts.s);
      image_store();                        // K rows j0..
      if (jc > 0) {                         // accumulated under the previous K block's scale
        const float ratio = kinv * isc.s;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) qacc[ft] *= ratio;
      }
      kinv = isc.inv;
      image_prefetch(Qb, i0, i0 + ni);      // next image: Q rows of this row block
      lds_barrier();
      DFGNN_DSTAMP(5)
      if constexpr (kBlocked) {
        // wave w: the dS rows of strips 2 (w / 2), 2 (w / 2) + 1 (from the tile: every strip put its own there before
        // the barrier) against the feature tiles of half w % 2 of K
        constexpr int NFT = FT / 2;
        const int cs0 = 2 * (wave >> 1), ft0 = NFT * (wave & 1);
        if (cs0 * 16 < ni) {
          const LaneIds L = lane_ids();
          f32x4 q0[NFT], q1[NFT];
#pragma unroll
          for (int k = 0; k < NFT; ++k) q0[k] = q1[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          const h16 *srow = Tb + (cs0 * 16 + L.mi) * TB + 8 * L.mq;
#pragma unroll
          for (int jb = 0; jb < CW / 32; ++jb) {
            if (32 * jb < nj) {
              const hx8 sh0 = *reinterpret_cast<const hx8 *>(srow + 32 * jb), sl0 = *reinterpret_cast<const hx8 *>(srow + TS + 32 * jb);
              const hx8 sh1 = *reinterpret_cast<const hx8 *>(srow + 16 * TB + 32 * jb),
                        sl1 = *reinterpret_cast<const hx8 *>(srow + 16 * TB + TS + 32 * jb);
              dense_kblock_mma2<NFT>(q0, q1, ihi, ilo, (32 * jb + 8 * L.mq + L.tq) * RS + 4 * L.tp + 16 * ft0, 4 * RS, sh0, sl0,
                                     sh1, sl1);
            }
          }
          dense_store_rows<NFT>(q0, kinv * ts.inv, dQb + 16 * ft0, (unsigned)hf, i0 + cs0 * 16 + L.mi, i0 + ni, L);
          dense_store_rows<NFT>(q1, kinv * ts.inv, dQb + 16 * ft0, (unsigned)hf, i0 + cs0 * 16 + 16 + L.mi, i0 + ni, L);
        }
      } else if (row_wave) {
        const LaneIds L = lane_ids();
        const h16 *srow = Tb + (wave * 16 + L.mi) * TB + 8 * L.mq;
#pragma unroll
        for (int jb = 0; jb < CW / 32; ++jb) {
          if (32 * jb < nj) {
            // natural k order: element t of lane (mi, mq) is column 32 jb + 8 mq + t of dS / that row of K
            const hx8 sh = *reinterpret_cast<const hx8 *>(srow + 32 * jb);
            const hx8 sl = *reinterpret_cast<const hx8 *>(srow + TS + 32 * jb);
            dense_kblock_mma<F, (NBLK == 1 ? 8 : 4)>(qacc, ihi, ilo, (32 * jb + 8 * L.mq + L.tq) * RS + 4 * L.tp, 4 * RS, sh, sl);
          }
        }
        if (jc + 1 == NBLK) {  // dQ rows of this row block are complete: store them now, under the dK product
          const int i = i0 + wave * 16 + L.mi;
          if constexpr (FR == F) dense_store_rows<FT>(qacc, kinv * ts.inv, dQb, (unsigned)hf, i, i0 + ni, L);
          else if (i < i0 + ni)
            dense_store_acc<FT, true>(qacc, kinv * ts.inv, dQb, (unsigned)i * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
        }
      }
      DFGNN_DSTAMP(6)
      image_post();     // Q rows of this row block
      lds_barrier();    // K image free
      image_store();
      const float dkscale = isc.inv * ts.inv;
      if (jc + 1 < NBLK) {
        image_prefetch(Kb, (jc + 1) * CW, n);
      } else if (rb + 1 < NRB && i0 + RB < n) {  // the next row block starts with its edges and its dO rows
        if constexpr (RECOMP) {
          image_prefetch(Kb, 0, n);  // ... or, with P recomputed, with the K rows of column block 0
        } else {
          ea = eb;
          eb = (i0 + 2 * RB < n) ? g.row_ptr[n0 + i0 + 2 * RB] : e0 + ne;
          image_prefetch(dOb, i0 + RB, min(n, i0 + 2 * RB));
        }
      }
      lds_barrier();
      DFGNN_DSTAMP(7)
      column_phase(dKb, j0, ni, !first, dkscale);
      if (jc + 1 < NBLK) image_post();  // K rows of the next column block
      lds_barrier();  // Q image and dS tile free
    }
    DFGNN_DSTAMP(8)
  }
  if constexpr (GAT) {
    const int tid = opaque_tid();
    if (tid < n) ga.grad_col[(size_t)(n0 + tid) * g.h + head] = gcol;
  }
}

}  // namespace dfgnn
