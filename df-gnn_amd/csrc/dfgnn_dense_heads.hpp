// dfgnn_dense_heads.hpp -- matrix-core GT forward for MULTI-HEAD configurations (h > 1, head width 16 / 32 / 64): one
// workgroup per dense range takes every head of the range.
//
// The per-(range, head) form (dense_fwd_body with MULTI) stages a K image and a V image per head -- for a 16-wide head
// that is a 128 x 16 slice zero-padded onto the 32-wide layout, 64-byte row segments from memory, five workgroup
// barriers and two scale reductions per head -- although everything after the logits (the masked softmax, the attention
// output, the P V product) is per-wave work that needs no barrier at all.  Here the heads are taken in GROUPS of 64
// feature columns (4 / 2 / 1 heads):
//   * the group's K columns and V columns are BOTH resident (two 64-wide fp16 hi / lo images, 256-byte row segments from
//     memory, the next group's travelling in registers): two barriers per group, none per head;
//   * a head's logits use the k-step of the K image that holds its columns (16-wide heads: the other head's half of the
//     k-step is zeroed in the Q operand); its P V product touches only its own 16-feature tiles of the V image;
//   * the byte map of the range lives in registers for all heads; the attention values of a strip are staged in a
//     per-wave LDS area and streamed out as whole lines; the group's 64 output columns of a row are stored together.
// Same numerics and operand layouts as dfgnn_dense.hpp.  Replaces, for such ranges, the same reference kernel as
// gt_dense.hip's forward (fused_gtconv_hyper.cu:228-560, one launch covers all heads there too: blockIdx.y).
#pragma once
#include "dfgnn_dense.hpp"
#include "dfgnn_dense_stamp.hpp"

namespace dfgnn {

constexpr int kHeadsGroupWidth = 64;

// head widths / head counts this body takes (the others stay on dense_fwd_body)
__host__ __device__ constexpr bool dense_heads_ok(int fr, int h) {
  return h > 1 && (fr == 16 || fr == 32 || fr == 64) && (h * fr) % kHeadsGroupWidth == 0;
}

// NFT feature tiles starting at element offset xoff of a 32-deep k-block: acc[k] += X^T Y (see dense_kblock_mma); a single
// tile keeps the three partial products in separate accumulators so that its MFMAs do not wait for each other
template <int NFT>
__device__ __forceinline__ void dense_kblock_mma_n(f32x4 (&acc)[NFT], f32x4 (&aux)[2], const h16 *ihi, const h16 *ilo,
                                                   int xoff, int second, const hx8 &yh, const hx8 &yl) {
  hx8 xh[NFT], xl[NFT];
#pragma unroll
  for (int k = 0; k < NFT; ++k) {
    xh[k] = dense_tr_pair(ihi + xoff + 16 * k, second);
    xl[k] = dense_tr_pair(ilo + xoff + 16 * k, second);
  }
  if constexpr (NFT == 1) {
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[0], yh, acc[0], 0, 0, 0);
    aux[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[0], yh, aux[0], 0, 0, 0);
    aux[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[0], yl, aux[1], 0, 0, 0);
  } else {
#pragma unroll
    for (int k = 0; k < NFT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yh, acc[k], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NFT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[k], yh, acc[k], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NFT; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yl, acc[k], 0, 0, 0);
  }
}

// FR: head width; NS strips per wave, NP padded rows (128 / 160); n <= NP, n > 8 * 16 * (NS - 1)
// STATS: the forward of the statistics-saving training pair (see dense_fwd_body): edge bitmaps instead of the byte map,
// row statistics (stat_max, stat_sum: [m, h]) instead of the attention values.
template <int FR, bool WRITE_ATTN, int NS, int NP, bool STATS = false>
__device__ __forceinline__ void dense_fwd_heads_body(float *lds, int lds_bytes, const Csr &g, int n0, int n, int e0, int ne,
                                                     const float *__restrict__ Q, const float *__restrict__ K,
                                                     const float *__restrict__ V, float *__restrict__ attn_edge,
                                                     float *__restrict__ out, float *__restrict__ stat_max = nullptr,
                                                     float *__restrict__ stat_sum = nullptr) {
  // STATS && WRITE_ATTN: the rank-ordered training forward (see dense_fwd_body): the map words hold RANKS -- the position of a
  // pair among the set bits of its row's bitmap -- and the values go out at row_ptr[i] + rank
  constexpr int FW = kHeadsGroupWidth, G = FW / FR, NT = NP / 16, KTH = FR == 64 ? 2 : 1, FTH = FR / 16;
  using D = DenseCfg<FW>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT;
  // the next group's images travel in registers while the current group is computed -- when the registers allow it
  // (one strip per wave); with two strips they are fetched at the group boundary
  constexpr bool kPrefetch = NS == 1;
  const int ngroups = g.h * FR / FW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int npad = (n + 31) & ~31, ntile = npad >> 4, nstrip = (n + 15) >> 4;
  const int MS = npad + 4;
  h16 *khi = reinterpret_cast<h16 *>(lds), *klo = khi + (size_t)NP * RS, *vhi = klo + (size_t)NP * RS, *vlo = vhi + (size_t)NP * RS;
  unsigned char *map = reinterpret_cast<unsigned char *>(vlo + (size_t)NP * RS);
  const int map_bytes = STATS ? 0 : nstrip * 16 * MS;
  int *rp = reinterpret_cast<int *>(map + ((map_bytes + 15) & ~15));
  float *smax = reinterpret_cast<float *>(rp + ((n + 4) & ~3));  // [8] per-wave maxima of the K image, [8] of the V image
  float *stage0 = smax + 2 * kDenseWaves;
  // per-wave staging area of the attention values of one strip: cap floats + one dump word per lane
  const int fixed_bytes = (int)(reinterpret_cast<char *>(stage0) - reinterpret_cast<char *>(lds));
  const int cap = (((lds_bytes - fixed_bytes) / kDenseWaves) / 4 - kWave) & ~3;
  float *wstage = stage0 + wave * (max(cap, 0) + kWave);
  DFGNN_LDS_AT(lds, (unsigned)fixed_bytes + (WRITE_ATTN ? (unsigned)(kDenseWaves * (max(cap, 0) + kWave) * 4) : 0u));  // the carve-up fits
  const size_t hf = (size_t)g.h * FR;
  const float *Qb = Q + (size_t)n0 * hf, *Kb = K + (size_t)n0 * hf, *Vb = V + (size_t)n0 * hf;
  float *Ob = out + (size_t)n0 * hf;

  DFGNN_DSTAMP(0)
  // ---- prologue: every long-latency load goes out before the first barrier, the small ones first -----------------------
  int rp_mine = 0;
  if constexpr (!STATS) {
    const int tid = opaque_tid();
    if (tid <= n) rp_mine = g.row_ptr[n0 + tid];
  }
  unsigned pre_c[kDensePre];
  if constexpr (!STATS) {
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < kDensePre; ++k) {
      const unsigned e = (unsigned)min(tid + k * kDenseThreads, ne - 1);
      pre_c[k] = ld32(g.coords + e0, e);
    }
  }
  DenseStageRegs<FW, NP> stK, stV;
  dense_stage_load<FW, NP>(stK, Kb, hf, 0, n);
  dense_stage_load<FW, NP>(stV, Vb, hf, 0, n);
  float4 qa[NS][KT], qb[NS][KT];  // this lane's pieces of its strips' Q rows, the 64 columns of one group, raw
  auto q_fetch = [&](const float *Qgroup) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      const unsigned off = (unsigned)min((wave + 8 * s) * 16 + L.mi, n - 1) * (unsigned)hf + 8u * L.mq;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        qa[s][t] = ld32_f4(Qgroup, off + 32 * t);
        qb[s][t] = ld32_f4(Qgroup, off + 32 * t + 4);
      }
    }
  };
  q_fetch(Qb);
  // STATS: the edge bitmaps of this lane's rows (plan.hip: masks), expanded below into the byte-map words the head loop
  // tests (0x00 = edge, 0xFF = none), so that both forms share it
  unsigned mbits[NS][STATS ? (NT + 1) / 2 : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      const unsigned *mp = g.mask + (size_t)(n0 + min((wave + 8 * s) * 16 + L.mi, n - 1)) * kPlanMaskWords;
#pragma unroll
      for (int w = 0; w < (NT + 1) / 2; ++w) mbits[s][w] = ld32(mp, (unsigned)w);
    }
  }
  if constexpr (!STATS) {
    const int tid = opaque_tid();
    for (int k = tid; k < (map_bytes >> 2); k += kDenseThreads) reinterpret_cast<unsigned *>(map)[k] = 0xFFFFFFFFu;
    if (tid <= n) rp[tid] = rp_mine - e0;
    lds_barrier();
  }
  if constexpr (!STATS) {  // byte map: position of every edge within its row (the plan guarantees distinct columns and rows < 255 long)
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < kDensePre; ++k) {
      const int e = tid + k * kDenseThreads;
      if (e < ne) {
        const int i = pre_c[k] >> 8, j = pre_c[k] & 0xFF;
        map[i * MS + j] = (unsigned char)(e - rp[i]);
      }
    }
    for (int e = tid + kDensePre * kDenseThreads; e < ne; e += kDenseThreads) {
      const unsigned c = g.coords[e0 + e];
      const int i = c >> 8, j = c & 0xFF;
      map[i * MS + j] = (unsigned char)(e - rp[i]);
    }
  }
  Pow2Scale ksc{1.f, 1.f}, vsc{1.f, 1.f};
  // group images: registers -> LDS around a barrier pair; the next group's are requested in between
  auto images_commit = [&](int next_group) {
    wg_max_post(smax, dense_stage_absmax<FW, NP>(stK));
    wg_max_post(smax + kDenseWaves, dense_stage_absmax<FW, NP>(stV));
    lds_barrier();  // (every strip is done with the previous group's images; the map is complete)
    ksc = pow2_scale(wg_max_read(smax));
    vsc = pow2_scale(wg_max_read(smax + kDenseWaves));
    dense_stage_store<FW, NP>(stK, khi, klo, ksc.s);
    dense_stage_store<FW, NP>(stV, vhi, vlo, vsc.s);
    if (kPrefetch && next_group < ngroups) {
      dense_stage_load<FW, NP>(stK, Kb + next_group * FW, hf, 0, n);
      dense_stage_load<FW, NP>(stV, Vb + next_group * FW, hf, 0, n);
    }
    lds_barrier();
  };
  images_commit(1);
  DFGNN_DSTAMP(1)
  // the strips' rows of the byte map, for every head
  unsigned mw[NS][NT];
  int row_e0[NS], strip_e0[NS], strip_e1[NS];  // rp of this lane's row; edge range of the strip
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const LaneIds L = lane_ids();
    const int strip = wave + 8 * s, i = strip * 16 + L.mi;
    const unsigned char *mrow = map + min(i, nstrip * 16 - 1) * MS + 4 * L.mq;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      if constexpr (STATS && WRITE_ATTN) {
        const unsigned word = mbits[s][STATS ? jt / 2 : 0], low = 16u * (jt & 1) + 4u * L.mq;
        const unsigned b4 = (i < n && jt < ntile) ? (word >> low) & 0xFu : 0u;
        unsigned before = 0;  // set bits of the row's earlier words
#pragma unroll
        for (int v = 0; v < jt / 2; ++v) before += __popc(mbits[s][STATS ? v : 0]);
        const unsigned r0 = before + __popc(word & ((1u << low) - 1u));
        const unsigned r1 = r0 + (b4 & 1u), r2 = r1 + ((b4 >> 1) & 1u), r3 = r2 + ((b4 >> 2) & 1u);
        mw[s][jt] = ((b4 & 1u) ? r0 : 0xFFu) | (((b4 & 2u) ? r1 : 0xFFu) << 8) | (((b4 & 4u) ? r2 : 0xFFu) << 16) |
                    (((b4 & 8u) ? r3 : 0xFFu) << 24);
      } else if constexpr (STATS) {
        const unsigned b = (i < n && jt < ntile) ? (mbits[s][STATS ? jt / 2 : 0] >> (16 * (jt & 1) + 4 * L.mq)) & 0xFu : 0u;
        // bit r -> byte r: 0x00 where the bit is set, 0xFF where it is not
        mw[s][jt] = ~(((b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21)) * 0xFFu);
      } else {
        mw[s][jt] = (strip < nstrip && jt < ntile) ? *reinterpret_cast<const unsigned *>(mrow + 16 * jt) : 0xFFFFFFFFu;
      }
    }
    if constexpr (!STATS) {
      row_e0[s] = rp[min(i, n)];
      strip_e0[s] = rp[min(n, strip * 16)];
      strip_e1[s] = rp[min(n, strip * 16 + 16)];
    } else if constexpr (WRITE_ATTN) {  // (no row_ptr copy in LDS here: three loads per strip)
      row_e0[s] = g.row_ptr[n0 + min(i, n)] - e0;
      strip_e0[s] = g.row_ptr[n0 + min(n, strip * 16)] - e0;
      strip_e1[s] = g.row_ptr[n0 + min(n, strip * 16 + 16)] - e0;
    }
  }

  for (int gq = 0;; ++gq) {  // ---- one group of 64 feature columns (G heads) per trip --------------------------------
    f32x4 o[NS][FT];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int head = gq * G + q;
      constexpr int kStepsPerHead = KTH;
      const int t0 = (q * FR) / 32;  // first k-step of the head's columns in the group's images
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int strip = wave + 8 * s;
        const LaneIds L = lane_ids();
        // ---- this head's Q operand of the strip: fp16 halves under the strip's own power-of-two scale (done for a
        //      strip past the range as well: nothing here may sit under a branch, the loads below least of all) -------
        hx8 qh[kStepsPerHead], ql[kStepsPerHead];
        float qinv;
        {
          const bool valid = strip * 16 + L.mi < n && (FR >= 32 || (L.mq >> 1) == (q & 1));
          float4 xa[kStepsPerHead], xb[kStepsPerHead];
          float qm = 0.f;
#pragma unroll
          for (int t = 0; t < kStepsPerHead; ++t) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            xa[t] = valid ? qa[s][t0 + t] : z;
            xb[t] = valid ? qb[s][t0 + t] : z;
            qm = fmaxf(qm, absmax8(xa[t], xb[t]));
          }
          const Pow2Scale qs = pow2_scale(wave_max(qm));
          qinv = qs.inv;
#pragma unroll
          for (int t = 0; t < kStepsPerHead; ++t) split_hx8(xa[t], xb[t], qs.s, qh[t], ql[t]);
        }
        // the last head of the group has converted its Q: the raw rows of the next group can be requested
        if (q == G - 1 && s == NS - 1 && gq + 1 < ngroups) q_fetch(Qb + (gq + 1) * FW);
        if (gq == 0 && q == 0 && s == 0) { DFGNN_DSTAMP(2) }
        if (strip < nstrip) {
          // One strip per wave: the compiler takes everything that depends only on the map words -- edge predicates, slots,
          // staging addresses -- out of the head loop (they are the same for every head), a welcome saving.  With two strips
          // per wave that is 80 more live registers and spills: there the words are re-read behind an optimisation barrier.
          unsigned mwh[NT];
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
            mwh[jt] = mw[s][jt];
            if constexpr (NS > 1) asm volatile("" : "+v"(mwh[jt]));
          }
          // ---- S^T tiles of the strip ------------------------------------------------------------------------------
          f32x4 S[NT];
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (u < ntile) {
#pragma unroll
              for (int t = 0; t < kStepsPerHead; ++t) {
                const int off = (16 * u + L.mi) * RS + 8 * L.mq + 32 * (t0 + t);
                const hx8 ah = *reinterpret_cast<const hx8 *>(khi + off), al = *reinterpret_cast<const hx8 *>(klo + off);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, qh[t], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, ql[t], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qh[t], acc, 0, 0, 0);
              }
            }
            S[u] = acc;
          }
          if (gq == 0 && q == 0 && s == 0) { DFGNN_DSTAMP(3) }
          // ---- masked row softmax in registers: the logits are S x c (c a power of two), the exponent is taken base 2 -
          const float c2 = (ksc.inv * qinv) * 1.4426950408889634f;
          float mx = -INFINITY;
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool edge = ((mwh[jt] >> (8 * r)) & 0xFFu) != 0xFFu;
              const float x = edge ? S[jt][r] : -INFINITY;
              S[jt][r] = x;
              mx = fmaxf(mx, x);
            }
          mx = xor16_32_max(mx);
          const float base = (mx == -INFINITY) ? 0.f : mx * c2;
          float sum = 0.f;
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = __builtin_amdgcn_exp2f(fmaf(S[jt][r], c2, -base));  // 2^-inf = 0 for the masked pairs
              S[jt][r] = p;
              sum += p;
            }
          sum = xor16_32_sum(sum);
          const float inv = (sum != 0.f) ? 1.f / sum : 0.f;
          if constexpr (STATS) {
            const int i = strip * 16 + L.mi;
            if (stat_max && i < n && L.mq == 0) {
              stat_max[(size_t)(n0 + i) * g.h + head] = (mx == -INFINITY) ? -1e38f : mx * (ksc.inv * qinv);
              stat_sum[(size_t)(n0 + i) * g.h + head] = sum;
            }
          }
          if (gq == 0 && q == 0 && s == 0) { DFGNN_DSTAMP(4) }
          if constexpr (WRITE_ATTN) {
            float *dst = attn_edge + (size_t)head * g.nnz + e0;
            const int s0 = strip_e0[s], s1 = strip_e1[s];
            if (s1 - s0 <= cap) {  // (wave-uniform) through this wave's staging area, then out as whole lines
              float *lrow = wstage + (row_e0[s] - s0);
              float *dump = wstage + cap + (threadIdx.x & (kWave - 1));
#pragma unroll
              for (int jt = 0; jt < NT; ++jt) {
                if (jt < ntile) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const unsigned slot = (mwh[jt] >> (8 * r)) & 0xFFu;
                    float *d = (slot != 0xFFu) ? lrow + slot : dump;
                    *d = S[jt][r] * inv;
                  }
                }
              }
              wave_sync();
              {  // four lines per trip: the LDS reads of a trip are issued together
                const int l = (int)(threadIdx.x & (kWave - 1));
                for (int e = s0 + l; e < s1; e += 4 * kWave) {
                  float v[4];
#pragma unroll
                  for (int k = 0; k < 4; ++k) v[k] = wstage[min(e + k * kWave, s1 - 1) - s0];
#pragma unroll
                  for (int k = 0; k < 4; ++k)
                    if (e + k * kWave < s1) dst[e + k * kWave] = v[k];
                }
              }
              wave_sync();  // (the area is reused by the wave's next strip / head)
            } else {  // a strip with more edges than the area holds: straight from the registers
              float *grow = dst + row_e0[s];
#pragma unroll
              for (int jt = 0; jt < NT; ++jt) {
                if (jt < ntile) {
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const unsigned slot = (mwh[jt] >> (8 * r)) & 0xFFu;
                    if (slot != 0xFFu) grow[slot] = S[jt][r] * inv;
                  }
                }
              }
            }
          }
          if (gq == 0 && q == 0 && s == 0) { DFGNN_DSTAMP(5) }
          // ---- O^T tiles of this head = V^T P^T: its FTH feature tiles of the group's V image ------------------------
          f32x4 acc[FTH], aux[2];
#pragma unroll
          for (int k = 0; k < FTH; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          aux[0] = aux[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jb = 0; jb < NP / 32; ++jb) {
            if (2 * jb < ntile) {
              hx8 yh, yl;
              dense_split8(S[2 * jb], S[2 * jb + 1], kUnitScale, yh, yl);
              dense_kblock_mma_n<FTH>(acc, aux, vhi, vlo, (32 * jb + 4 * L.mq + L.tq) * RS + 4 * L.tp + 16 * (q * FTH), 16 * RS, yh, yl);
            }
          }
          const float oscale = inv * (vsc.inv * kUnitScaleInv);
#pragma unroll
          for (int k = 0; k < FTH; ++k) {
            if constexpr (FTH == 1) acc[k] += aux[0] + aux[1];
            o[s][q * FTH + k] = acc[k] * oscale;
          }
        } else {
#pragma unroll
          for (int k = 0; k < FTH; ++k) o[s][q * FTH + k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (gq == 0 && q == 0 && s == 0) { DFGNN_DSTAMP(6) }
      }
    }
    if (gq == 0) { DFGNN_DSTAMP(7) }
    // the group's 64 output columns of every row, as whole lines
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const LaneIds L = lane_ids();
      if ((wave + 8 * s) * 16 < n)
        dense_store_rows<FT>(o[s], 1.f, Ob + gq * FW, (unsigned)hf, (wave + 8 * s) * 16 + L.mi, n, L);
    }
    if (gq == 0) { DFGNN_DSTAMP(8) }
    if (gq + 1 >= ngroups) break;
    if constexpr (!kPrefetch) {
      dense_stage_load<FW, NP>(stK, Kb + (gq + 1) * FW, hf, 0, n);
      dense_stage_load<FW, NP>(stV, Vb + (gq + 1) * FW, hf, 0, n);
    }
    images_commit(gq + 2);
    if (gq == 0) { DFGNN_DSTAMP(9) }
  }
  DFGNN_DSTAMP(10)
}

}  // namespace dfgnn

namespace dfgnn {

// =====================================================================================================================
// backward, every head of a range of <= 128 nodes in one workgroup
// =====================================================================================================================
// The per-(range, head) backward (gt_dense.hip: dense_bwd_body) starts cold for every head -- edge coordinates,
// attention values and the dO image have to arrive before anything can be done (a quarter of its time for 16-wide
// heads) -- and swaps one image buffer four times per head, a barrier pair each.  Here a workgroup walks the heads of
// its range:
//   * the edge coordinates are fetched once (registers) and reused by every head's scatter;
//   * TWO image buffers: dO and V are resident together (dV^T = dO^T P and dP = dO V^T need no barrier between them),
//     then K and Q (dQ = dS K and dK^T = Q^T dS likewise); four barriers per head;
//   * the next pair of images and the next head's attention values travel in registers while the current pair is used;
//   * the tile is never cleared after the first head: dS is exactly zero where P is, so the pairs that are no edge
//     already hold zeros when the next head's P values are scattered onto the (identical) edge positions.
// P and dS live in the tile as interleaved fp16 hi | lo rows (dfgnn_dense_wide.hpp's form: P under the constant scale
// 2^14, dS under the tile's power-of-two scale).  Same numerics and operand layouts as dfgnn_dense.hpp.
// Replaces, for such ranges, fused_gtconv_backward.cu:40-191 (which is single-head only).
// dense_stage_store / dense_rows_mma_strip with the image row stride as a parameter (the 160-row walk uses a tighter
// one than dfgnn_dense.hpp's F + 16 to fit two images next to its tile)
template <int F, int ROWS, int RS>
__device__ __forceinline__ void dense_stage_store_rs(DenseStageRegs<F, ROWS> &r, h16 *hi, h16 *lo, float scale, int fr) {
  constexpr int C8 = F / 8;
  const int tid = opaque_tid();
#pragma unroll
  for (int k = 0; k < DenseStageRegs<F, ROWS>::PER; ++k) {
    const int idx = tid + k * kDenseThreads;
    const int row = idx / C8, c8 = idx - row * C8;
    if (idx < ROWS * C8) {
      const bool valid = r.row0 + row < r.row_end && (fr >= F || 8 * c8 < fr);
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      hx8 h, l;
      split_hx8(valid ? r.a[k] : z, valid ? r.b[k] : z, scale, h, l);
      *reinterpret_cast<hx8 *>(hi + row * RS + 8 * c8) = h;
      *reinterpret_cast<hx8 *>(lo + row * RS + 8 * c8) = l;
    }
  }
  asm volatile("" ::: "memory");
}
template <int F, int NTILES, int RS>
__device__ __forceinline__ void dense_rows_mma_strip_rs(f32x4 (&out)[NTILES], const h16 *ihi, const h16 *ilo, int limit,
                                                        const hx8 (&xh)[F / 32], const hx8 (&xl)[F / 32], const LaneIds &L) {
#pragma unroll
  for (int u = 0; u < NTILES; ++u) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (16 * u < limit) {
      const int off = (16 * u + L.mi) * RS + 8 * L.mq;
#pragma unroll
      for (int t = 0; t < F / 32; ++t) {
        const hx8 ah = *reinterpret_cast<const hx8 *>(ihi + off + 32 * t), al = *reinterpret_cast<const hx8 *>(ilo + off + 32 * t);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[t], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[t], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[t], acc, 0, 0, 0);
      }
    }
    out[u] = acc;
  }
}

template <int FR, int NP = 128>
__device__ __forceinline__ void dense_bwd_heads_body(float *lds, const Csr &g, int n0, int n, int e0, int ne, int head0,
                                                     int nheads, const float *__restrict__ Q, const float *__restrict__ K,
                                                     const float *__restrict__ V, const float *__restrict__ attn_edge,
                                                     const float *__restrict__ dO, float *__restrict__ dQ,
                                                     float *__restrict__ dK, float *__restrict__ dV) {
  constexpr int F = FR < 32 ? 32 : FR, U = NP / 16, NS = (U + kDenseWaves - 1) / kDenseWaves, PRE = kDensePre;
  constexpr int fr = FR;
  using D = DenseCfg<F>;
  // image row stride: F + 16 elements (conflict-free row and transposed reads) when two images fit next to the tile that
  // way, else F + 8 (160 rows: some two-way conflicts in the transposed reads, but no third and fourth barrier pair)
  constexpr int RS = (NP > 128) ? F + 8 : D::RS, KT = D::KT, FT = D::FT, TS = NP + 8, TB = 2 * TS;
  static_assert(NP == 128 || F == 32, "the 160-row walk is for heads of at most 32 features");
  static_assert((4 * NP * RS + NP * TB) * 2 + 3 * kDenseWaves * 4 <= kLdsBytes, "LDS");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int nstrip = (n + 15) >> 4;
  h16 *ahi = reinterpret_cast<h16 *>(lds), *alo = ahi + (size_t)NP * RS;   // buffer A: dO, then K
  h16 *bhi = alo + (size_t)NP * RS, *blo = bhi + (size_t)NP * RS;         // buffer B: V, then Q
  h16 *Tb = blo + (size_t)NP * RS;                                        // the tile: NP rows of hi[TS] | lo[TS]
  float *smax = reinterpret_cast<float *>(Tb + (size_t)NP * TB);          // [3][8] per-wave maxima: image A, image B, dS
  DFGNN_LDS_AT(lds, (unsigned)(reinterpret_cast<char *>(smax + 3 * kDenseWaves) - reinterpret_cast<char *>(lds)));  // the carve-up fits
  const size_t hf = (size_t)g.h * fr;
  const float *Qb = Q + (size_t)n0 * hf, *Kb = K + (size_t)n0 * hf, *Vb = V + (size_t)n0 * hf, *dOb = dO + (size_t)n0 * hf;
  float *dQb = dQ + (size_t)n0 * hf, *dKb = dK + (size_t)n0 * hf, *dVb = dV + (size_t)n0 * hf;
  const float *attn0 = attn_edge + e0;
  const int hend = head0 + nheads;  // this workgroup walks the heads head0 .. hend - 1

  // ---- prologue: coordinates (kept), the first head's attention values, dO and V images -------------------------------
  unsigned pc[PRE];
  float pa[PRE];
  auto attn_fetch = [&](const float *attn_h) {
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PRE; ++k) pa[k] = ld32(attn_h, (unsigned)min(tid + k * kDenseThreads, max(ne, 1) - 1));
  };
  {
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PRE; ++k) pc[k] = ld32(g.coords + e0, (unsigned)min(tid + k * kDenseThreads, max(ne, 1) - 1));
  }
  attn_fetch(attn0 + (size_t)head0 * g.nnz);
  DenseStageRegs<F, NP> stA, stB;
  dense_stage_load<F, NP>(stA, dOb + head0 * fr, hf, 0, n, fr);
  dense_stage_load<F, NP>(stB, Vb + head0 * fr, hf, 0, n, fr);
  {
    const int tid = opaque_tid();
    for (int k = tid; k < NP * TB / 8; k += kDenseThreads) reinterpret_cast<float4 *>(Tb)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  wg_max_post(smax, dense_stage_absmax<F, NP>(stA));
  wg_max_post(smax + kDenseWaves, dense_stage_absmax<F, NP>(stB));
  lds_barrier();

  // P of every edge -> the tile, fp16 hi | lo under the scale 2^14 (the edges past the prefetched ones from memory)
  auto scatter = [&](const float *attn_h) {
    const int tid = opaque_tid();
    auto put = [&](unsigned c, float p) {
      const int at = (int)(c >> 8) * TB + (int)(c & 0xFF);
      const h16 hh = (h16)(p * kUnitScale);
      Tb[at] = hh;
      Tb[at + TS] = (h16)fmaf(p, kUnitScale, -(float)hh);
    };
#pragma unroll
    for (int k = 0; k < PRE; ++k)
      if (tid + k * kDenseThreads < ne) put(pc[k], pa[k]);
    for (int e = tid + PRE * kDenseThreads; e < ne; e += kDenseThreads) put(g.coords[e0 + e], attn_h[e]);
  };
  // out^T[f][c] = sum_i X[i][f] Y[i][c]: X = an image, Y = the tile.  Column strip cs, all FT feature tiles (ft < 0: the
  // tile fragments are shared) or the single feature tile ft
  auto column_unit = [&](const h16 *xhi, const h16 *xlo, float *outb, int cs, int ft, float oscale) {
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
          dense_kblock_mma<F, 4>(acc, xhi, xlo, (32 * ib + 4 * L.mq + L.tq) * RS + 4 * L.tp, 16 * RS, yh, yl);
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
          const hx8 xh = dense_tr_pair(xhi + xoff, 16 * RS);
          const hx8 xl = dense_tr_pair(xlo + xoff, 16 * RS);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yh, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, yh, acc[0], 0, 0, 0);
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, yl, acc[0], 0, 0, 0);
        }
      }
      if (j < n)
        dense_store_acc<1, (FR < F)>(acc, oscale, outb, (unsigned)j * (unsigned)hf + 16u * ft + 4u * L.mq, false, 16 * ft + 4 * L.mq, fr);
    }
  };
  // all column strips of the range: the first eight whole, the others dealt out tile by tile so that all waves share them
  auto column_phase = [&](const h16 *xhi, const h16 *xlo, float *outb, float oscale) {
    if (wave < nstrip) column_unit(xhi, xlo, outb, wave, -1, oscale);
    if constexpr (NS > 1)
      for (int unit = wave; unit < (nstrip - kDenseWaves) * FT; unit += kDenseWaves)
        column_unit(xhi, xlo, outb, kDenseWaves + unit / FT, unit % FT, oscale);
  };

  for (int hd = head0;; ++hd) {  // ---- one head per trip --------------------------------------------------------------------
    const unsigned hoff = (unsigned)hd * fr;
    // images: dO -> A, V -> B (their maxima were posted ahead of the last barrier); tile := P
    const Pow2Scale dosc = pow2_scale(wg_max_read(smax)), vsc = pow2_scale(wg_max_read(smax + kDenseWaves));
    dense_stage_store_rs<F, NP, RS>(stA, ahi, alo, dosc.s, fr);
    dense_stage_store_rs<F, NP, RS>(stB, bhi, blo, vsc.s, fr);
    dense_stage_load<F, NP>(stA, Kb + hoff, hf, 0, n, fr);  // the next pair: K, Q of this head
    dense_stage_load<F, NP>(stB, Qb + hoff, hf, 0, n, fr);
    scatter(attn0 + (size_t)hd * g.nnz);
    lds_barrier();  // B0
    if (hd + 1 < hend) attn_fetch(attn0 + (size_t)(hd + 1) * g.nnz);  // (the values of this head are in the tile)

    // ---- dP^T = V dO^T ; t ; dS (registers) ; dV^T = dO^T P ----------------------------------------------------------
    f32x4 dS[NS][U];
    float tmax = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int strip = wave + kDenseWaves * s;
      if (strip < nstrip) {
        const LaneIds L = lane_ids();
        hx8 gh[KT], gl[KT];  // this strip's dO rows: the register operand of dP (the image's scale)
        const int off = (strip * 16 + L.mi) * RS + 8 * L.mq;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          gh[t] = *reinterpret_cast<const hx8 *>(ahi + off + 32 * t);
          gl[t] = *reinterpret_cast<const hx8 *>(alo + off + 32 * t);
        }
        f32x4 Pr[U];
        const h16 *trow = Tb + (strip * 16 + L.mi) * TB + 4 * L.mq;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const hx4 h4 = *reinterpret_cast<const hx4 *>(trow + 16 * u), l4 = *reinterpret_cast<const hx4 *>(trow + TS + 16 * u);
#pragma unroll
          for (int r = 0; r < 4; ++r) Pr[u][r] = ((float)h4[r] + (float)l4[r]) * kUnitScaleInv;
        }
        dense_rows_mma_strip_rs<F, U, RS>(dS[s], bhi, blo, n, gh, gl, L);  // dP (x the two scales)
        const float dpinv = vsc.inv * dosc.inv;
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dS[s][u][r] *= dpinv;
            t = fmaf(Pr[u][r], dS[s][u][r], t);
          }
        t = xor16_32_sum(t);  // a row lives on 4 lanes of this wave
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dS[s][u][r] = Pr[u][r] * (dS[s][u][r] - t);
            tmax = fmaxf(tmax, fabsf(dS[s][u][r]));
          }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) dS[s][u] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    column_phase(ahi, alo, dVb + hoff, dosc.inv * kUnitScaleInv);
    wg_max_post(smax, dense_stage_absmax<F, NP>(stA));                 // K
    wg_max_post(smax + kDenseWaves, dense_stage_absmax<F, NP>(stB));   // Q
    wg_max_post(smax + 2 * kDenseWaves, tmax);
    lds_barrier();  // B1: every wave is done with the P tile and the dO / V images

    // ---- tile := dS ; images: K -> A, Q -> B ; dQ = dS K ; dK^T = Q^T dS ----------------------------------------------
    const Pow2Scale ksc = pow2_scale(wg_max_read(smax)), qsc = pow2_scale(wg_max_read(smax + kDenseWaves));
    const Pow2Scale ts = pow2_scale(wg_max_read(smax + 2 * kDenseWaves));
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
            const h16 hh = (h16)(dS[s][u][r] * ts.s);
            h4[r] = hh;
            l4[r] = (h16)fmaf(dS[s][u][r], ts.s, -(float)hh);
          }
          *reinterpret_cast<hx4 *>(trow + 16 * u) = h4;
          *reinterpret_cast<hx4 *>(trow + TS + 16 * u) = l4;
        }
      }
    }
    dense_stage_store_rs<F, NP, RS>(stA, ahi, alo, ksc.s, fr);
    dense_stage_store_rs<F, NP, RS>(stB, bhi, blo, qsc.s, fr);
    const bool more = hd + 1 < hend;
    {  // the next head's dO and V (after the last head: one clamped row each, cache hits, never stored)
      const unsigned noff = more ? hoff + fr : hoff;
      dense_stage_load<F, NP>(stA, dOb + noff, hf, 0, more ? n : 1, fr);
      dense_stage_load<F, NP>(stB, Vb + noff, hf, 0, more ? n : 1, fr);
    }
    lds_barrier();  // B2
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int strip = wave + kDenseWaves * s;
      if (strip < nstrip) {
        const LaneIds L = lane_ids();
        f32x4 qacc[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) qacc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        const h16 *srow = Tb + (strip * 16 + L.mi) * TB + 8 * L.mq;
#pragma unroll
        for (int jb = 0; jb < NP / 32; ++jb) {
          if (32 * jb < n) {
            // natural k order: element t of lane (mi, mq) is column 32 jb + 8 mq + t of dS / that row of K
            const hx8 sh = *reinterpret_cast<const hx8 *>(srow + 32 * jb);
            const hx8 sl = *reinterpret_cast<const hx8 *>(srow + TS + 32 * jb);
            dense_kblock_mma<F, 4>(qacc, ahi, alo, (32 * jb + 8 * L.mq + L.tq) * RS + 4 * L.tp, 4 * RS, sh, sl);
          }
        }
        const int i = strip * 16 + L.mi;
        if constexpr (FR == F) dense_store_rows<FT>(qacc, ksc.inv * ts.inv, dQb + hoff, (unsigned)hf, i, n, L);
        else if (i < n) dense_store_acc<FT, true>(qacc, ksc.inv * ts.inv, dQb + hoff, (unsigned)i * (unsigned)hf + 4u * L.mq, false, 4 * L.mq, fr);
      }
    }
    column_phase(bhi, blo, dKb + hoff, qsc.inv * ts.inv);
    if (!more) break;
    wg_max_post(smax, dense_stage_absmax<F, NP>(stA));                 // dO of the next head
    wg_max_post(smax + kDenseWaves, dense_stage_absmax<F, NP>(stB));   // V
    lds_barrier();  // B3: every wave is done with the dS tile and the K / Q images
  }
}

}  // namespace dfgnn
