// dfgnn_dense_lean.hpp -- matrix-core GT forward for dense ranges of <= 128 nodes, TWO workgroups per CU.
//
// The 512-thread forward (gt_dense.hip: dense_fwd_body) owns a CU: 8 waves x up to 256 VGPRs and ~115 KB of LDS.  Its
// phases alternate between moving bytes (prologue: edges + K + Q; V image) and computing (S, softmax, P V), and a CU
// streams from HBM at most its share of the chip's rate (~10 B / cycle: tools/diag/stream_probe.hip reaches 6.2 TB/s
// only with every CU loading ALL the time), so the memory pipe of a CU idles while its one workgroup computes: the
// forward runs at ~55 % of the streaming rate.  This body is sized so that TWO workgroups fit a CU -- 256 threads
// (4 waves, each owning strips w and w + 4), <= 256 VGPRs, < 80 KB of LDS -- and the hardware overlaps one workgroup's
// loads with the other's matrix / vector phases:
//   * feature matrices are staged 64 columns at a time (40 KB image): S accumulates over the two K halves, the two V
//     halves give the two halves of the output row;
//   * the next half image travels in registers while the current one is used (as in the 512-thread kernels).
// Same numerics and layouts as dfgnn_dense.hpp (fp16 hi / lo operand halves under power-of-two scales).
// Replaces, for such ranges, the same reference kernels as gt_dense.hip (fused_gtconv_hyper.cu:165-532).
#pragma once
#include "dfgnn_dense.hpp"

namespace dfgnn {

constexpr int kLeanThreads = 256;
constexpr int kLeanWaves = kLeanThreads / kWave;
constexpr int kLeanLdsBytes = 80 * 1024;

// STATS: the forward of the statistics-saving training pair (see dense_fwd_body): edge bitmaps instead of the byte map,
// row statistics (stat_max, stat_sum: [m, h]) instead of the attention values.
template <int F, bool WRITE_ATTN, bool STATS = false>
__global__ __launch_bounds__(kLeanThreads, 2) void gt_dense_fwd_lean_kernel(Csr g, const int *__restrict__ fit,
                                                                           const float *__restrict__ Q,
                                                                           const float *__restrict__ K,
                                                                           const float *__restrict__ V,
                                                                           float *__restrict__ attn_edge,
                                                                           float *__restrict__ out,
                                                                           float *__restrict__ stat_max = nullptr,
                                                                           float *__restrict__ stat_sum = nullptr) {
  static_assert(F == 64 || F == 128, "lean forward: whole 64-column halves only");
  // STATS && WRITE_ATTN: the rank-ordered training forward (see dense_fwd_body)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int FW = 64, NH = F / FW, NP = 128, NT = NP / 16, NS = 2, PRE = 16;
  using D = DenseCfg<FW>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT, MS = NP + 4;
  constexpr int PER = NP * (FW / 8) / kLeanThreads;  // 8-float pieces of an image per thread (4)
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0, head = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int npad = (n + 31) & ~31, ntile = npad >> 4, nstrip = (n + 15) >> 4;
  h16 *ihi = reinterpret_cast<h16 *>(lds), *ilo = ihi + NP * RS;
  unsigned char *map = reinterpret_cast<unsigned char *>(ilo + NP * RS);
  const int map_bytes = STATS ? 0 : nstrip * 16 * MS;
  int *rp = reinterpret_cast<int *>(map + ((map_bytes + 15) & ~15));
  float *smax = reinterpret_cast<float *>(rp + ((n + 4) & ~3));  // [4] per-wave maxima of the image being staged
  float *pstage = smax + 4;                                       // [ne] normalised attention values, if it fits
  const size_t fixed_bytes = (size_t)(reinterpret_cast<char *>(pstage) - reinterpret_cast<char *>(lds));
  const bool stage_attn = WRITE_ATTN && fixed_bytes + (size_t)ne * 4 <= (size_t)kLeanLdsBytes;
  DFGNN_LDS_AT(lds, (unsigned)(fixed_bytes + (stage_attn ? (size_t)ne * 4 : 0)));  // the carve-up fits
  const size_t hf = (size_t)g.h * F, hoff = (size_t)head * F;
  const float *Qb = Q + (size_t)n0 * hf + hoff, *Kb = K + (size_t)n0 * hf + hoff, *Vb = V + (size_t)n0 * hf + hoff;
  float *Ob = out + (size_t)n0 * hf + hoff;

  // ---- half images through registers ---------------------------------------------------------------------------------
  float4 ia[PER], ib[PER];
  auto image_fetch = [&](const float *src) {  // rows 0 .. n-1 (clamped), 64 columns starting at src
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int idx = tid + k * kLeanThreads;
      const unsigned off = (unsigned)min(idx >> 3, n - 1) * (unsigned)hf + 8u * (idx & 7);
      ia[k] = ld32_f4(src, off);
      ib[k] = ld32_f4(src, off + 4);
    }
  };
  auto image_post = [&]() {
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) mx = fmaxf(mx, absmax8(ia[k], ib[k]));
    mx = wave_max(mx);
    if ((threadIdx.x & (kWave - 1)) == 0) smax[wave] = mx;
  };
  Pow2Scale isc{1.f, 1.f};
  auto image_store = [&]() {
    const float4 w = *reinterpret_cast<const float4 *>(smax);
    isc = pow2_scale(fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w)));
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int idx = tid + k * kLeanThreads, row = idx >> 3, c8 = idx & 7;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      hx8 hh, ll;
      split_hx8(row < n ? ia[k] : z, row < n ? ib[k] : z, isc.s, hh, ll);
      *reinterpret_cast<hx8 *>(ihi + row * RS + 8 * c8) = hh;
      *reinterpret_cast<hx8 *>(ilo + row * RS + 8 * c8) = ll;
    }
    asm volatile("" ::: "memory");
  };
  // this wave's Q rows, one feature half: raw, then fp16 operand fragments under a per-strip scale
  float4 qa[NS][KT], qb[NS][KT];
  auto q_fetch = [&](int h) {
    const LaneIds L = lane_ids();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const unsigned off = (unsigned)min((wave + kLeanWaves * s) * 16 + L.mi, n - 1) * (unsigned)hf + (unsigned)(h * FW) + 8u * L.mq;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        qa[s][t] = ld32_f4(Qb, off + 32 * t);
        qb[s][t] = ld32_f4(Qb, off + 32 * t + 4);
      }
    }
  };
  hx8 qh[NS][KT], ql[NS][KT];
  float qinv[NS];
  auto q_convert = [&]() {
    const LaneIds L = lane_ids();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bool valid = (wave + kLeanWaves * s) * 16 + L.mi < n;
      float qm = 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        if (!valid) qa[s][t] = qb[s][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        qm = fmaxf(qm, absmax8(qa[s][t], qb[s][t]));
      }
      const Pow2Scale qs = pow2_scale(wave_max(qm));
      qinv[s] = qs.inv;
#pragma unroll
      for (int t = 0; t < KT; ++t) split_hx8(qa[s][t], qb[s][t], qs.s, qh[s][t], ql[s][t]);
    }
  };

  // ---- prologue: everything that is needed first is requested first ------------------------------------------------
  int rp_mine = 0;
  if constexpr (!STATS || WRITE_ATTN) {
    const int tid = opaque_tid();
    if (tid <= n) rp_mine = g.row_ptr[n0 + tid];
  }
  unsigned pre_c[PRE];  // packed (row, column) within the range (plan.hip: coords)
  if constexpr (!STATS) {
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const unsigned e = (unsigned)min(tid + k * kLeanThreads, ne - 1);
      pre_c[k] = ld32(g.coords + e0, e);
    }
  }
  image_fetch(Kb);
  q_fetch(0);
  unsigned mwords[NS][STATS ? NT / 2 : 1];  // STATS: the edge bitmaps of this lane's rows (plan.hip: masks)
  if constexpr (STATS) {
    const LaneIds L = lane_ids();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint4 w = *reinterpret_cast<const uint4 *>(
          g.mask + (size_t)(n0 + min((wave + kLeanWaves * s) * 16 + L.mi, n - 1)) * kPlanMaskWords);
      mwords[s][0] = w.x; mwords[s][1] = w.y; mwords[s][2] = w.z; mwords[s][3] = w.w;
    }
  }
  if constexpr (STATS && WRITE_ATTN) {  // (read long after the image barriers below)
    const int tid = opaque_tid();
    if (tid <= n) rp[tid] = rp_mine - e0;
  }
  if constexpr (!STATS) {
    const int tid = opaque_tid();
    for (int k = tid; k < (map_bytes >> 2); k += kLeanThreads) reinterpret_cast<unsigned *>(map)[k] = 0xFFFFFFFFu;
    if (tid <= n) rp[tid] = rp_mine - e0;
    lds_barrier();
  }
  if constexpr (!STATS) {  // byte map: position of every edge within its row (0xFF: no edge)
    const int tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int e = tid + k * kLeanThreads;
      if (e < ne) {
        const int i = pre_c[k] >> 8, j = pre_c[k] & 0xFF;
        map[i * MS + j] = (unsigned char)(e - rp[i]);
      }
    }
    constexpr int B = 8;
    for (int base = PRE * kLeanThreads; base < ne; base += B * kLeanThreads) {
      unsigned bc[B];
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const unsigned e = (unsigned)min(base + tid + k * kLeanThreads, ne - 1);
        bc[k] = ld32(g.coords + e0, e);
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const int e = base + tid + k * kLeanThreads;
        if (e < ne) map[(bc[k] >> 8) * MS + (bc[k] & 0xFF)] = (unsigned char)(e - rp[bc[k] >> 8]);
      }
    }
  }
  image_post();
  lds_barrier();
  image_store();  // K, half 0
  q_convert();
  lds_barrier();

  // ---- S = Q K^T, accumulated over the feature halves ---------------------------------------------------------------
  f32x4 S[NS][NT];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int u = 0; u < NT; ++u) S[s][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    if (h + 1 < NH) {
      image_fetch(Kb + (h + 1) * FW);
      q_fetch(h + 1);
    } else {
      image_fetch(Vb);
    }
    {
      const LaneIds L = lane_ids();
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (wave + kLeanWaves * s < nstrip) {
          const float c = isc.inv * qinv[s];
#pragma unroll
          for (int u = 0; u < NT; ++u)
            if (u < ntile) S[s][u] += dense_rows_mma<FW>(ihi, ilo, u, qh[s], ql[s], L) * c;
        }
      }
    }
    if (h + 1 < NH) {
      image_post();
      lds_barrier();
      image_store();  // K, half h + 1
      q_convert();
      lds_barrier();
    }
  }

  // ---- masked row softmax, in registers ------------------------------------------------------------------------------
  float inv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    inv[s] = 0.f;
    const int strip = wave + kLeanWaves * s;
    if (strip < nstrip) {
      const LaneIds L = lane_ids();
      const int i = strip * 16 + L.mi;
      const unsigned char *mrow = map + i * MS + 4 * L.mq;
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        unsigned w = 0xFFFFFFFFu, wb = 0u;
        if constexpr (STATS) wb = (i < n && jt < ntile) ? (mwords[s][STATS ? jt / 2 : 0] >> (16 * (jt & 1) + 4 * L.mq)) & 0xFu : 0u;
        else w = (jt < ntile) ? *reinterpret_cast<const unsigned *>(mrow + 16 * jt) : 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool edge = STATS ? ((wb >> r) & 1u) != 0u : ((w >> (8 * r)) & 0xFFu) != 0xFFu;
          const float x = edge ? S[s][jt][r] : -INFINITY;
          S[s][jt][r] = x;
          mx = fmaxf(mx, x);
        }
      }
      mx = xor16_32_max(mx);
      const float base = (mx == -INFINITY) ? 0.f : mx;
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = fast_exp(S[s][jt][r] - base);  // exp(-inf) = 0 for the masked pairs
          S[s][jt][r] = p;
          sum += p;
        }
      sum = xor16_32_sum(sum);
      inv[s] = (sum != 0.f) ? 1.f / sum : 0.f;
      if constexpr (STATS) {
        if (stat_max && i < n && L.mq == 0) {
          stat_max[(size_t)(n0 + i) * g.h + head] = (mx == -INFINITY) ? -1e38f : mx;
          stat_sum[(size_t)(n0 + i) * g.h + head] = sum;
        }
      }
      if constexpr (WRITE_ATTN) {
        if (i < n) {
          float *lrow = pstage + rp[i];
          float *grow = attn_edge + (size_t)head * g.nnz + e0 + rp[i];
          auto scatter = [&](float *row) {
            unsigned before = 0;  // (STATS) set bits of the bitmap words in front of the current one
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
              if (jt < ntile) {
                unsigned w;
                if constexpr (!STATS) {
                  w = *reinterpret_cast<const unsigned *>(mrow + 16 * jt);
                } else {  // the ranks of this lane's 4 pairs among the set bits of its row's bitmap (see dense_fwd_body)
                  const unsigned word = mwords[s][STATS ? jt / 2 : 0], low = 16u * (jt & 1) + 4u * L.mq;
                  const unsigned b4 = (word >> low) & 0xFu;
                  const unsigned r0 = before + __popc(word & ((1u << low) - 1u));
                  const unsigned r1 = r0 + (b4 & 1u), r2 = r1 + ((b4 >> 1) & 1u), r3 = r2 + ((b4 >> 2) & 1u);
                  if (jt & 1) before += __popc(word);
                  w = ((b4 & 1u) ? r0 : 0xFFu) | (((b4 & 2u) ? r1 : 0xFFu) << 8) | (((b4 & 4u) ? r2 : 0xFFu) << 16) |
                      (((b4 & 8u) ? r3 : 0xFFu) << 24);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const unsigned slot = (w >> (8 * r)) & 0xFFu;
                  if (slot != 0xFFu) row[slot] = S[s][jt][r] * inv[s];
                }
              }
            }
          };
          if (stage_attn) scatter(lrow);
          else scatter(grow);
        }
        if (stage_attn) {  // the strip streams its own contiguous slice of attn_edge out
          wave_sync();
          const int s0 = rp[strip * 16], s1 = rp[min(n, strip * 16 + 16)];
          float *dst = attn_edge + (size_t)head * g.nnz + e0;
          for (int e = s0 + (int)(threadIdx.x & (kWave - 1)); e < s1; e += kWave) dst[e] = pstage[e];
        }
      }
    }
  }

  // ---- O = P V, one feature half at a time ---------------------------------------------------------------------------
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    image_post();
    lds_barrier();  // every strip is done with the previous image
    image_store();  // V, half h
    if (h + 1 < NH) image_fetch(Vb + (h + 1) * FW);
    lds_barrier();
    const LaneIds L = lane_ids();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int strip = wave + kLeanWaves * s;
      if (strip < nstrip) {
        f32x4 o[FT];
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) o[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jb = 0; jb < NP / 32; ++jb)
          if (2 * jb < ntile) dense_cols_mma<FW, 4>(o, ihi, ilo, jb, S[s][2 * jb], S[s][2 * jb + 1], kUnitScale, L);
        dense_store_rows<FT>(o, inv[s] * (isc.inv * kUnitScaleInv), Ob + h * FW, (unsigned)hf, strip * 16 + L.mi, n, L);
      }
    }
  }
}

}  // namespace dfgnn
