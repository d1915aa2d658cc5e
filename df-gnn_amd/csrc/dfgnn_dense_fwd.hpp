// dfgnn_dense_fwd.hpp -- the forward body of the matrix-core GT / GAT kernels (gt_dense.hip, gt_dense_stats.hip).
// See gt_dense.hip for the design, dfgnn_dense.hpp for the numerics and the operand layouts.
#pragma once
#include "dfgnn_dense.hpp"
#include "dfgnn_dense_stamp.hpp"

namespace dfgnn {

// =====================================================================================================================
// forward
// =====================================================================================================================
// NS strips per wave, chunks of CR rows of K / V, NCH chunks: (1, 128, 1) up to 128 nodes, (2, 160, 1) up to 160,
// (2, 128, 2) up to 255.
// GAT = true: the logits are LeakyReLU(attn_row[i] + attn_col[j]) instead of <Q_i, K_j> (Q = attn_row [m, h],
// K = attn_col [m, h], V = X): no K image and no first product, everything else is shared.
// Attention dropout of the GAT training pair (gat_train.hip: GatDrop): keep edge e of head hd iff
// mask[e * h + hd] > drop, kept attention scaled by `scale` = 1 / (1 - drop).  mask == NULL: no dropout.
struct DenseDrop {
  const float *mask = nullptr;
  float drop = 0.f, scale = 1.f;
};

// FR: the real feature width.  Widths below the narrowest MFMA k-step (f = 16: the heads of multi-head GT configs) run
// zero-padded on the 32-wide layout (F below); for FR >= 32 every padding guard folds away at compile time.
// MULTI: the workgroup loops over the heads of its range (GT, h > 1); false: one head, no loop (values that are live
// around a loop -- the prefetch registers -- would be spilled in the single-head kernel too)
// STATS (GT training forward of the statistics-saving pair, gt_dense_stats.hip): the edge set comes from the plan's
// bitmaps (g.mask: each lane fetches the words of its own rows straight from memory -- no byte map, no edge list, no
// row pointers), the normalised attention is not written, the row statistics (logit maximum, sum of exponentials) are.
template <int FR, bool WRITE_ATTN, int NS, int CR, int NCH, bool GAT = false, bool MULTI = false, bool STATS = false, bool WEIGHTED = false>
__device__ __forceinline__ void dense_fwd_body(float *lds, int lds_bytes, const Csr &g, int n0, int n, int e0, int ne,
                                               int head, int nheads, const float *__restrict__ Q,
                                               const float *__restrict__ K, const float *__restrict__ V,
                                               float *__restrict__ attn_edge,
                                               float *__restrict__ out, float slope = 0.f,
                                               float *__restrict__ stat_max = nullptr,
                                               float *__restrict__ stat_sum = nullptr,
                                               const DenseDrop drop = DenseDrop{}) {
  static_assert(!(STATS && GAT), "the bitmap-driven forward is a GT form");
  // STATS && WRITE_ATTN: the "ranked" training forward -- edges from the bitmaps, attention values written in RANK order
  // (row i's k-th edge by increasing column at row_ptr[i] + k; the slot of a pair = the number of set bits before it),
  // which the backward reads back with the plan's rank-ordered coordinates: no byte map, no edge loads, no scatter
  constexpr int F = FR < 32 ? 32 : FR;  // layout width
  constexpr int fr = FR;
  static_assert(!WEIGHTED || (STATS && !GAT), "dense edge values go with the bitmap-masked (STATS) GT forward");
  using D = DenseCfg<F>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT, TPC = CR / 16, NT = TPC * NCH;
  constexpr int MW = (NT + 1) / 2;  // bitmap words of a row (STATS)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int npad = (n + 31) & ~31, ntile = npad >> 4, nstrip = (n + 15) >> 4;
  const int MS = npad + 4;
  h16 *ihi = reinterpret_cast<h16 *>(lds), *ilo = ihi + (size_t)CR * RS;
  unsigned char *map = reinterpret_cast<unsigned char *>(ilo + (size_t)CR * RS);
  const int map_bytes = STATS ? 0 : nstrip * 16 * MS;
  int *rp = reinterpret_cast<int *>(map + ((map_bytes + 15) & ~15));
  float *smax = reinterpret_cast<float *>(rp + ((n + 4) & ~3));    // [8] per-wave maxima of the image being staged
  float *acl = smax + kDenseWaves;                                 // [npad] attn_col of the range (GAT only)
  float *pstage = acl + (GAT ? npad : 0);                          // [ne] normalised attention values, if it fits
  const size_t fixed_bytes = (size_t)(reinterpret_cast<char *>(pstage) - reinterpret_cast<char *>(lds));
  const bool stage_attn = WRITE_ATTN && fixed_bytes + ((size_t)ne + kDenseThreads) * 4 <= (size_t)lds_bytes;  // (+ dump words)
  DFGNN_LDS_AT(lds, (unsigned)(fixed_bytes + (stage_attn ? ((size_t)ne + kDenseThreads) * 4 : 0)));  // the carve-up fits
  // GT: the workgroup takes the heads head .. head + nheads - 1 of its range one after the other -- the edge loads and
  // the byte map are shared, the next head's K image and Q rows travel while the current head's P V product runs
  // (GAT: nheads = 1; its attn_col staging and dropout map are per head)
  const size_t hf = (size_t)g.h * fr, hoff = (size_t)head * fr;
  const float *Qb = Q + (size_t)n0 * hf + hoff, *Kb = K + (size_t)n0 * hf + hoff, *Vb = V + (size_t)n0 * hf + hoff;
  float *Ob = out + (size_t)n0 * hf + hoff;
  const int hend = MULTI ? head + nheads : head + 1;
  // the next head's K image and Q rows are fetched under the current head's P V product for narrow heads only: wide ones
  // have the registers for neither (they would spill around the loop) nor the need (a head is long)
  constexpr bool kNextHeadPrefetch = MULTI && F <= 32;
  (void)Qb;
  (void)Kb;

  DFGNN_DSTAMP(0)
  // ---- every long-latency load of the prologue goes out before the first barrier; the small ones that are needed
  //      first go first (memory returns in order: behind the big loads they would wait for all of them) ---------------
  int rp_mine = 0;       // row_ptr[n0 + tid] (n <= 255: one entry per thread covers the range)
  float ac_mine = 0.f;   // GAT: attn_col[n0 + tid]
  if constexpr (!STATS || WRITE_ATTN) {
    const int tid = opaque_tid();
    if (tid <= n) rp_mine = g.row_ptr[n0 + tid];
    if constexpr (GAT)
      if (tid < n) ac_mine = K[(size_t)(n0 + tid) * g.h + head];
  }
  unsigned pre_c[kDensePre];         // packed (row, column) of the edges within the range (plan.hip: coords)
  float pre_m[GAT ? kDensePre : 1];  // GAT with dropout: the edges' uniform randoms
  const float *mask_h = nullptr;     // ... of this head, edge e at mask_h[(e0 + e) * h]
  if constexpr (GAT)
    if (drop.mask) mask_h = drop.mask + (size_t)e0 * g.h + head;
  if constexpr (!STATS) {
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < kDensePre; ++k) {
      const unsigned e = (unsigned)min(tid + k * kDenseThreads, ne - 1);  // clamped: plain loads
      pre_c[k] = ld32(g.coords + e0, e);
      if constexpr (GAT) pre_m[k] = mask_h ? mask_h[(size_t)e * g.h] : 1.f;
    }
  }
  DenseStageRegs<F, CR> st;
  dense_stage_load<F, CR>(st, GAT ? Vb : Kb, hf, 0, n, fr);  // the first image: K rows (GAT: X rows)
  float4 qa[NS][KT], qb[NS][KT];  // this lane's pieces of its strips' Q rows, raw: converted after the map is built
  float ar[NS];
  auto q_fetch = [&](const float *Qhead) {  // (GT) the strips' Q rows of one head
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      const int i = min((wave + 8 * s) * 16 + L.mi, n - 1);
      const unsigned off = (unsigned)i * (unsigned)hf;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const unsigned c = (FR == F || 32 * t + 8 * L.mq < fr) ? 32u * t + 8u * L.mq : 0u;  // (past fr: zeroed below)
        qa[s][t] = ld32_f4(Qhead, off + c);
        qb[s][t] = ld32_f4(Qhead, off + c + 4);
      }
    }
  };
  if constexpr (GAT) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      ar[s] = Q[(size_t)(n0 + min((wave + 8 * s) * 16 + L.mi, n - 1)) * g.h + head];
    }
  } else {
    q_fetch(Qb);
  }
  unsigned mwords[NS][STATS ? MW : 1];  // STATS: the edge bitmaps of this lane's rows (rows past the range: no edges)
  if constexpr (STATS) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      const int i = (wave + 8 * s) * 16 + L.mi;
      const unsigned *mp = g.mask + (size_t)(n0 + min(i, n - 1)) * kPlanMaskWords;
#pragma unroll
      for (int w = 0; w < MW; ++w) mwords[s][w] = ld32(mp, (unsigned)w);
    }
  }
  if constexpr (STATS && WRITE_ATTN) {  // (read long after the image barriers below)
    const int tid = opaque_tid();
    if (tid <= n) rp[tid] = rp_mine - e0;
  }
  if constexpr (!STATS) {
    const int tid = opaque_tid();
    for (int k = tid; k < (map_bytes >> 2); k += kDenseThreads) reinterpret_cast<unsigned *>(map)[k] = 0xFFFFFFFFu;
    if (tid <= n) rp[tid] = rp_mine - e0;
    if constexpr (GAT)
      if (tid < npad) acl[tid] = ac_mine;
    lds_barrier();
  }
  if constexpr (!STATS) {  // byte map: position of every edge within its row (the plan guarantees distinct columns and rows < 255 long);
     // the edge loads were issued first, so this runs while the K rows are still on their way.
     // GAT with dropout (positions are not needed there): 0 = kept edge, 1 = dropped edge.
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < kDensePre; ++k) {
      const int e = tid + k * kDenseThreads;
      if (e < ne) {
        const int i = pre_c[k] >> 8, j = pre_c[k] & 0xFF;
        if (GAT && mask_h) map[i * MS + j] = (pre_m[GAT ? k : 0] > drop.drop) ? 0 : 1;
        else map[i * MS + j] = (unsigned char)(e - rp[i]);
      }
    }
    for (int e = tid + kDensePre * kDenseThreads; e < ne; e += kDenseThreads) {
      const unsigned c = g.coords[e0 + e];
      const int i = c >> 8, j = c & 0xFF;
      if (GAT && mask_h) map[i * MS + j] = (mask_h[(size_t)e * g.h] > drop.drop) ? 0 : 1;
      else map[i * MS + j] = (unsigned char)(e - rp[i]);
    }
  }
  for (int hd = head;; ++hd) {  // ---- one head of the range per trip (MULTI) ----------------------------------------
  // The image's power-of-two scale needs the largest magnitude over the whole workgroup: one more barrier here (the
  // later images post theirs ahead of a barrier that is there anyway).
  wg_max_post(smax, dense_stage_absmax<F, CR>(st));
  lds_barrier();
  Pow2Scale isc = pow2_scale(wg_max_read(smax));  // scale of the resident image
  dense_stage_store<F, CR>(st, ihi, ilo, isc.s, fr);
  float kinv[NCH];    // 1 / scale of K chunk c ...
  float qinv[NS];     // ... and of this wave's Q strips: S = acc * kinv * qinv
  kinv[0] = isc.inv;
  hx8 qh[NS][KT], ql[NS][KT];
#pragma unroll
  for (int s = 0; s < NS; ++s) qinv[s] = 1.f;
  if constexpr (!GAT) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      const bool valid = (wave + 8 * s) * 16 + L.mi < n;
      float qm = 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        if (!valid || (FR < F && 32 * t + 8 * L.mq >= fr)) qa[s][t] = qb[s][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        qm = fmaxf(qm, absmax8(qa[s][t], qb[s][t]));
      }
      const Pow2Scale qs = pow2_scale(wave_max(qm));
      qinv[s] = qs.inv;
#pragma unroll
      for (int t = 0; t < KT; ++t) split_hx8(qa[s][t], qb[s][t], qs.s, qh[s][t], ql[s][t]);
    }
  }
  lds_barrier();
  DFGNN_DSTAMP(1)
  // the next image (the second K chunk of a two-chunk range, else V rows 0..) lands during the S phase
  if constexpr (GAT) {
    if (NCH > 1) dense_stage_load<F, CR>(st, Vb, hf, CR, n, fr);
  } else {
    if (NCH == 1) dense_stage_load<F, CR>(st, Vb, hf, 0, n, fr);
    else dense_stage_load<F, CR>(st, Kb, hf, CR, n, fr);
  }

  // ---- S^T = K Q^T (GAT: the rank-one logits) ------------------------------------------------------------------------
  f32x4 S[NS][NT];
  if constexpr (GAT) {
    const LaneIds L = lane_ids();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const float4 a = (jt < ntile) ? *reinterpret_cast<const float4 *>(acl + 16 * jt + 4 * L.mq) : make_float4(0.f, 0.f, 0.f, 0.f);
        S[s][jt] = f32x4{leaky_relu(ar[s] + a.x, slope), leaky_relu(ar[s] + a.y, slope), leaky_relu(ar[s] + a.z, slope),
                         leaky_relu(ar[s] + a.w, slope)};
      }
  }
#pragma unroll
  for (int c = 0; c < (GAT ? 0 : NCH); ++c) {
    if (c > 0) {
      wg_max_post(smax, dense_stage_absmax<F, CR>(st));
      lds_barrier();
      isc = pow2_scale(wg_max_read(smax));
      kinv[c] = isc.inv;
      dense_stage_store<F, CR>(st, ihi, ilo, isc.s, fr);
      dense_stage_load<F, CR>(st, Vb, hf, 0, n, fr);  // V rows 0.., for the first O^T chunk
      lds_barrier();
    }
    const LaneIds L = lane_ids();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (wave + 8 * s < nstrip) {
        if constexpr (NS == 1 && NCH == 1) {
          dense_rows_mma_strip<F, TPC>(S[s], ihi, ilo, 16 * ntile, qh[s], ql[s], L);  // (double-buffered fragments)
        } else {
#pragma unroll
          for (int u = 0; u < TPC; ++u) {
            const int jt = TPC * c + u;
            S[s][jt] = (jt < ntile) ? dense_rows_mma<F>(ihi, ilo, u, qh[s], ql[s], L) : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      }
    }
  }
  DFGNN_DSTAMP(2)

  // ---- masked row softmax, in registers -------------------------------------------------------------------------------
  float inv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    inv[s] = 0.f;
    const int strip = wave + 8 * s;
    if (strip < nstrip) {
      const LaneIds L = lane_ids();
      const int i = strip * 16 + L.mi;
      const unsigned char *mrow = map + i * MS + 4 * L.mq;
      // STATS: the 4 pairs of tile jt are the bits 16 (jt & 1) + 4 mq .. + 3 of word jt / 2 of the row's bitmap
      auto stat_bits = [&](int jt) -> unsigned {
        return (i < n && jt < ntile) ? (mwords[s][STATS ? jt / 2 : 0] >> (16 * (jt & 1) + 4 * L.mq)) & 0xFu : 0u;
      };
      // One K chunk: the accumulators are the logits up to ONE positive factor (the two power-of-two scales), so the
      // row maximum is taken on them as they are and the factor -- times log2 e -- goes into the exponent's FMA:
      // p = 2^(S c2 - max c2).  Two chunks have a scale each: the logits are formed first.
      constexpr bool kFold = NCH == 1;
      // WEIGHTED: the logit of an edge is S val (the reference's attn * val, fused_gtconv_hyper.cu:88-90), val from the
      // dense weight rows of the plan; the common positive factor of the accumulators still folds into the exponent
      const float *wrow = WEIGHTED ? g.wdense + (size_t)(n0 + min(i, n - 1)) * kPlanWeightStride + 4 * L.mq : nullptr;
      const float c2 = (GAT ? 1.f : kinv[0] * qinv[s]) * 1.4426950408889634f;
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        unsigned w = 0xFFFFFFFFu, wb = 0u;
        if constexpr (STATS) wb = stat_bits(jt);
        else w = (jt < ntile) ? *reinterpret_cast<const unsigned *>(mrow + 16 * jt) : 0xFFFFFFFFu;
        float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);
        if constexpr (WEIGHTED)
          if (jt < ntile) wv = ld32_f4(wrow, 16u * jt);
        const float wvr[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool edge = STATS ? ((wb >> r) & 1u) != 0u : ((w >> (8 * r)) & 0xFFu) != 0xFFu;
          float x = edge ? ((GAT || kFold) ? S[s][jt][r] : S[s][jt][r] * (kinv[jt / TPC] * qinv[s])) : -INFINITY;
          if constexpr (WEIGHTED) x = edge ? x * wvr[r] : x;
          S[s][jt][r] = x;
          mx = fmaxf(mx, x);
        }
      }
      mx = xor16_32_max(mx);
      if constexpr (kFold && !GAT) mx = (mx == -INFINITY) ? mx : mx * (kinv[0] * qinv[s]);  // the logit maximum itself
      const float base = (mx == -INFINITY) ? 0.f : mx;
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // exp(-inf) = 0 for the masked pairs
          const float p = kFold ? __builtin_amdgcn_exp2f(fmaf(S[s][jt][r], c2, -base * 1.4426950408889634f)) : fast_exp(S[s][jt][r] - base);
          S[s][jt][r] = p;
          sum += p;
        }
      sum = xor16_32_sum(sum);
      inv[s] = (sum != 0.f) ? 1.f / sum : 0.f;
      if constexpr (GAT || STATS) {  // training forward: the row statistics the backward recomputes P from
        if (stat_max && i < n && L.mq == 0) {
          stat_max[(size_t)(n0 + i) * g.h + hd] = (mx == -INFINITY) ? -1e38f : mx;
          stat_sum[(size_t)(n0 + i) * g.h + hd] = sum;
        }
      }
      if constexpr (GAT) {
        if (mask_h) {  // attention dropout after the softmax: the row sum counted every edge, the product skips
          inv[s] *= drop.scale;  // the dropped ones
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            const unsigned w = (jt < ntile) ? *reinterpret_cast<const unsigned *>(mrow + 16 * jt) : 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (((w >> (8 * r)) & 0xFFu) == 1u) S[s][jt][r] = 0.f;
          }
        }
      }
      if constexpr (WRITE_ATTN) {
        // attn_edge (CSR order): through LDS when the range's edge array fits (then the strip streams its own
        // contiguous slice out), else straight from the registers (scattered 4-byte stores)
        if (i < n) {
          float *lrow = pstage + rp[i];
          float *grow = attn_edge + (size_t)hd * g.nnz + e0 + rp[i];
          // the 4 slots of tile jt as one word, 0xFF = no edge: from the byte map, or (STATS) the ranks of the pairs among
          // the set bits of the row's bitmap -- bits before this lane's 4 within the word + the words before it
          unsigned before = 0;  // (STATS) set bits of the words in front of the current one
          auto slots_of = [&](int jt) -> unsigned {
            if constexpr (!STATS) {
              return *reinterpret_cast<const unsigned *>(mrow + 16 * jt);
            } else {
              const unsigned word = mwords[s][jt / 2], low = 16u * (jt & 1) + 4u * L.mq;
              const unsigned b4 = (word >> low) & 0xFu;
              const unsigned r0 = before + __popc(word & ((1u << low) - 1u));
              const unsigned r1 = r0 + (b4 & 1u), r2 = r1 + ((b4 >> 1) & 1u), r3 = r2 + ((b4 >> 2) & 1u);
              if (jt & 1) before += __popc(word);
              return ((b4 & 1u) ? r0 : 0xFFu) | (((b4 & 2u) ? r1 : 0xFFu) << 8) | (((b4 & 4u) ? r2 : 0xFFu) << 16) |
                     (((b4 & 8u) ? r3 : 0xFFu) << 24);
            }
          };
          if (stage_attn) {
            // LDS staging, branch-free: a pair that is no edge writes into this lane's own dump word instead of being
            // masked out (an exec-mask round trip per pair costs more than the store)
            float *dump = pstage + ne + (threadIdx.x & (kDenseThreads - 1));
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
              if (jt < ntile) {
                const unsigned w = slots_of(jt);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const unsigned slot = (w >> (8 * r)) & 0xFFu;
                  float *dst = (slot != 0xFFu) ? lrow + slot : dump;
                  *dst = S[s][jt][r] * inv[s];
                }
              }
            }
          } else {
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
              if (jt < ntile) {
                const unsigned w = slots_of(jt);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const unsigned slot = (w >> (8 * r)) & 0xFFu;
                  if (slot != 0xFFu) grow[slot] = S[s][jt][r] * inv[s];
                }
              }
            }
          }
        }
        if (stage_attn) {
          wave_sync();
          const int s0 = rp[strip * 16], s1 = rp[min(n, strip * 16 + 16)];
          float *dst = attn_edge + (size_t)hd * g.nnz + e0;
          // four lines per trip: the LDS reads of a trip are issued together (one read -> one store per trip is a chain of
          // LDS round trips: 13 of them for a strip of 830 edges)
          for (int e = s0 + (int)(threadIdx.x & (kWave - 1)); e < s1; e += 4 * kWave) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = pstage[min(e + k * kWave, s1 - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (e + k * kWave < s1) dst[e + k * kWave] = v[k];
          }
        }
      }
    }
  }
  DFGNN_DSTAMP(3)

  // ---- O^T = V^T P^T --------------------------------------------------------------------------------------------------
  f32x4 o[NS][FT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int ft = 0; ft < FT; ++ft) o[s][ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  // P (the exp values, in [0, 1]) enters the product under the constant scale 2^14, V under its image's; the
  // accumulators are kept in units of the current image's scale.
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (!(GAT && c == 0)) {  // (GAT: X rows 0.. are the image already)
      wg_max_post(smax, dense_stage_absmax<F, CR>(st));
      lds_barrier();       // every strip is done with the previous image
      const float prev_inv = isc.inv;
      isc = pow2_scale(wg_max_read(smax));
      dense_stage_store<F, CR>(st, ihi, ilo, isc.s, fr);
      if (c + 1 < NCH) {
        dense_stage_load<F, CR>(st, Vb, hf, (c + 1) * CR, n, fr);
      } else if (kNextHeadPrefetch && hd + 1 < hend) {  // the next head's first image and Q rows
        dense_stage_load<F, CR>(st, Kb + fr, hf, 0, n, fr);
        q_fetch(Qb + fr);
      }
      if (c > 0) {  // accumulated under the previous chunk's scale
        const float ratio = prev_inv * isc.s;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int ft = 0; ft < FT; ++ft) o[s][ft] *= ratio;
      }
      lds_barrier();
    }
    if (c == 0) { DFGNN_DSTAMP(4) }
    const LaneIds L = lane_ids();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (wave + 8 * s < nstrip) {
#pragma unroll
        for (int u = 0; u < CR / 32; ++u) {
          const int jb = (CR / 32) * c + u;  // 32-column block of P
          if (2 * jb < ntile)
            dense_cols_mma<F, (NS == 1 ? 8 : 4)>(o[s], ihi, ilo, u, S[s][2 * jb], S[s][2 * jb + 1], kUnitScale, L);
        }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const LaneIds L = lane_ids();
    const int i = (wave + 8 * s) * 16 + L.mi;
    if constexpr (FR == F) {
      if ((wave + 8 * s) * 16 < n) dense_store_rows<FT>(o[s], inv[s] * (isc.inv * kUnitScaleInv), Ob, (unsigned)hf, i, n, L);
    } else if (i < n) {
      dense_store_acc<FT, true>(o[s], inv[s] * (isc.inv * kUnitScaleInv), Ob, (unsigned)i * (unsigned)hf + 4u * L.mq, false,
                                4 * L.mq, fr);
    }
  }
  DFGNN_DSTAMP(5)
  if (!MULTI || hd + 1 >= hend) break;
  Qb += fr; Kb += fr; Vb += fr; Ob += fr;
  if constexpr (!kNextHeadPrefetch) {
    dense_stage_load<F, CR>(st, Kb, hf, 0, n, fr);
    q_fetch(Qb);
  }
  }  // (heads)
  DFGNN_DSTAMP(6)
}

}  // namespace dfgnn
