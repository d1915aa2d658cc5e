// gt_dense.hip -- matrix-core GT forward and backward for the dense ranges of a block plan.
//
// One workgroup of 8 waves per dense range (single-head; multi-head: per range for the forward and for the backward of
// ranges of <= 128 nodes, which walk the heads -- dfgnn_dense_heads.hpp -- else per (range, head)); wave w owns the
// 16-row strip w (and strip w + 8 of a range with more than 128 nodes).  The sparse structure of a range arrives as
// one uint16 per edge (row << 8 | column inside the range: the plan's `coords`, plan.hip).  See dfgnn_dense.hpp for the
// numerics and the operand layouts.  These kernels replace,
// for such ranges, the same reference kernels as gt_block.hip / gt_block_bwd.hip (fused_gtconv_hyper.cu:228-560,
// fused_gtconv_backward.cu:40-191): the dot products, the softmax and the weighted sums of a whole member graph
// are done as masked dense attention on v_mfma_f32_16x16x32_f16 (fp16 hi / lo operand halves under power-of-two
// scales: fp32-equivalent, see dfgnn_dense.hpp).
//
// Forward:   S^T = K Q^T  ->  masked row softmax in registers  ->  O^T = V^T P^T
//            The mask is a byte map [i][j] -> position of edge (i, j) in row i (0xFF: no edge), built once per
//            range in LDS from the CSR arrays; it also tells where P_ij goes in attn_edge.
// Backward:  P (attn_edge) is scattered into a dense tile (one-tile GT ranges: directly as fp16 hi | lo halves; else fp32,
//            converted in place by the strips); off-edge pairs have P = 0, hence dS = 0: no mask.
//            dV^T = dO^T P first (dO image resident; the strips take their dO rows -- the register operand of the next
//            product -- from it, so dO is read from global memory once), then
//            dP^T = V dO^T ;  t_i = sum_j P_ij dP_ij ;  dS = P o (dP - t)        (registers)
//            dQ^T = K^T dS^T ,  dK^T = Q^T dS                                    (P / dS through an fp16 tile in LDS)
//            A row block of 128 rows keeps dP / P / dS of all (one or two) 128-column blocks in registers, so t_i
//            needs no extra sweep and dQ accumulates in registers; dK / dV of a two-block range are accumulated
//            across the row blocks by the lane that wrote them.
// LDS: one feature image (K then V; dO, V, K, Q in turn; 72 KB) + the byte map (forward) or one 128 x 136-float
// tile (backward: P as fp32, then P and dS as interleaved fp16 hi | lo rows, 68 KB).  The next image's global
// loads are issued one phase ahead into registers, so a phase change costs a barrier and an LDS store.  Every barrier
// is lds_barrier() (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also wait for vmcnt(0), i.e. for the
// image that was just prefetched -- no thread ever reads another thread's global writes in these kernels.
#include <cstdlib>
#include <type_traits>

#include "dfgnn_dense.hpp"

namespace dfgnn {

#ifdef DFGNN_STAMPS
__device__ unsigned long long *dfgnn_dense_stamps = nullptr;  // [wg][16] phase boundaries (diagnostic build only)
#define DFGNN_DSTAMP(k)                                                                             \
  if (threadIdx.x == 0 && dfgnn_dense_stamps)                                                       \
    dfgnn_dense_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memtime();
// per-workgroup trace (same diagnostic build): [wg][8] = entry time, exit time (s_memtime), hardware id
// (XCC_ID << 32 | HW_ID: which CU ran it), entry and exit time on the constant 100 MHz clock
__device__ unsigned long long *dfgnn_wg_trace = nullptr;
#define DFGNN_TRACE_IN                                                                              \
  if (threadIdx.x == 0 && dfgnn_wg_trace) {                                                         \
    unsigned long long *t_ = dfgnn_wg_trace + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8;   \
    t_[0] = __builtin_amdgcn_s_memtime();                                                           \
    t_[3] = __builtin_amdgcn_s_memrealtime();                                                       \
    t_[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); \
  }
#define DFGNN_TRACE_OUT                                                                             \
  if (threadIdx.x == 0 && dfgnn_wg_trace) {                                                         \
    unsigned long long *t_ = dfgnn_wg_trace + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8;   \
    t_[1] = __builtin_amdgcn_s_memtime();                                                           \
    t_[4] = __builtin_amdgcn_s_memrealtime();                                                       \
  }
#else
#define DFGNN_DSTAMP(k)
#define DFGNN_TRACE_IN
#define DFGNN_TRACE_OUT
#endif

}  // namespace dfgnn
#ifndef DFGNN_FW128
#define DFGNN_FW128 128
#endif
#ifndef DFGNN_RING160
#define DFGNN_RING160 2  // prefetch distance (image phases) of the 129..160-node backward
#endif
#include "dfgnn_dense_wide.hpp"
#include "dfgnn_dense_lean.hpp"
#include "dfgnn_dense_heads.hpp"
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
template <int FR, bool WRITE_ATTN, int NS, int CR, int NCH, bool GAT = false, bool MULTI = false>
__device__ __forceinline__ void dense_fwd_body(float *lds, int lds_bytes, const Csr &g, int n0, int n, int e0, int ne,
                                               int head, int nheads, const float *__restrict__ Q,
                                               const float *__restrict__ K, const float *__restrict__ V,
                                               float *__restrict__ attn_edge,
                                               float *__restrict__ out, float slope = 0.f,
                                               float *__restrict__ stat_max = nullptr,
                                               float *__restrict__ stat_sum = nullptr,
                                               const DenseDrop drop = DenseDrop{}) {
  constexpr int F = FR < 32 ? 32 : FR;  // layout width
  constexpr int fr = FR;
  using D = DenseCfg<F>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT, TPC = CR / 16, NT = TPC * NCH;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int npad = (n + 31) & ~31, ntile = npad >> 4, nstrip = (n + 15) >> 4;
  const int MS = npad + 4;
  h16 *ihi = reinterpret_cast<h16 *>(lds), *ilo = ihi + (size_t)CR * RS;
  unsigned char *map = reinterpret_cast<unsigned char *>(ilo + (size_t)CR * RS);
  const int map_bytes = nstrip * 16 * MS;
  int *rp = reinterpret_cast<int *>(map + ((map_bytes + 15) & ~15));
  float *smax = reinterpret_cast<float *>(rp + ((n + 4) & ~3));    // [8] per-wave maxima of the image being staged
  float *acl = smax + kDenseWaves;                                 // [npad] attn_col of the range (GAT only)
  float *pstage = acl + (GAT ? npad : 0);                          // [ne] normalised attention values, if it fits
  const size_t fixed_bytes = (size_t)(reinterpret_cast<char *>(pstage) - reinterpret_cast<char *>(lds));
  const bool stage_attn = WRITE_ATTN && fixed_bytes + ((size_t)ne + kDenseThreads) * 4 <= (size_t)lds_bytes;  // (+ dump words)
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
  {
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
  {
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
  {
    const int tid = opaque_tid();
    for (int k = tid; k < (map_bytes >> 2); k += kDenseThreads) reinterpret_cast<unsigned *>(map)[k] = 0xFFFFFFFFu;
    if (tid <= n) rp[tid] = rp_mine - e0;
    if constexpr (GAT)
      if (tid < npad) acl[tid] = ac_mine;
  }
  lds_barrier();
  {  // byte map: position of every edge within its row (the plan guarantees distinct columns and rows < 255 long);
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
      // One K chunk: the accumulators are the logits up to ONE positive factor (the two power-of-two scales), so the
      // row maximum is taken on them as they are and the factor -- times log2 e -- goes into the exponent's FMA:
      // p = 2^(S c2 - max c2).  Two chunks have a scale each: the logits are formed first.
      constexpr bool kFold = NCH == 1;
      const float c2 = (GAT ? 1.f : kinv[0] * qinv[s]) * 1.4426950408889634f;
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const unsigned w = (jt < ntile) ? *reinterpret_cast<const unsigned *>(mrow + 16 * jt) : 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool edge = ((w >> (8 * r)) & 0xFFu) != 0xFFu;
          const float x = edge ? ((GAT || kFold) ? S[s][jt][r] : S[s][jt][r] * (kinv[jt / TPC] * qinv[s])) : -INFINITY;
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
      if constexpr (GAT) {  // training forward: the row statistics the backward recomputes P from
        if (stat_max && i < n && L.mq == 0) {
          stat_max[(size_t)(n0 + i) * g.h + hd] = (mx == -INFINITY) ? -1e38f : mx;
          stat_sum[(size_t)(n0 + i) * g.h + hd] = sum;
        }
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
          if (stage_attn) {
            // LDS staging, branch-free: a pair that is no edge writes into this lane's own dump word instead of being
            // masked out (an exec-mask round trip per pair costs more than the store)
            float *dump = pstage + ne + (threadIdx.x & (kDenseThreads - 1));
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
              if (jt < ntile) {
                const unsigned w = *reinterpret_cast<const unsigned *>(mrow + 16 * jt);
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
                const unsigned w = *reinterpret_cast<const unsigned *>(mrow + 16 * jt);
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
          for (int e = s0 + (int)(threadIdx.x & (kWave - 1)); e < s1; e += kWave) dst[e] = pstage[e];
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

template <int F, bool WRITE_ATTN>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_fwd_kernel(Csr g, const int *__restrict__ fit,
                                                                     const float *__restrict__ Q,
                                                                     const float *__restrict__ K,
                                                                     const float *__restrict__ V,
                                                                     float *__restrict__ attn_edge,
                                                                     float *__restrict__ out, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  DFGNN_TRACE_IN
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if (g.h == 1) {
    if (n <= kDenseChunkRows)
      dense_fwd_body<F, WRITE_ATTN, 1, kDenseChunkRows, 1>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_edge, out);
    else if (n <= kDenseWideRows)
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseWideRows, 1>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_edge, out);
    else
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseChunkRows, 2>(lds, lds_bytes, g, n0, n, e0, ne, 0, 1, Q, K, V, attn_edge, out);
  } else {  // every head of the range in this workgroup: edge loads and byte map once
    if constexpr (F == 16 || F == 32 || F == 64) {
      if (dense_heads_ok(F, g.h) && n <= kDenseWideRows) {  // heads in groups of 64 columns (dfgnn_dense_heads.hpp)
#ifdef DFGNN_HEADS_VARIANT  // diagnostic builds: one geometry only (register use of a single body)
        constexpr int hv = DFGNN_HEADS_VARIANT;
#else
        constexpr int hv = -1;
#endif
        if ((hv < 0 && n <= kDenseChunkRows) || hv == 0)
          dense_fwd_heads_body<F, WRITE_ATTN, 1, kDenseChunkRows>(lds, lds_bytes, g, n0, n, e0, ne, Q, K, V, attn_edge, out);
        else if (hv != 0)
          dense_fwd_heads_body<F, WRITE_ATTN, 2, kDenseWideRows>(lds, lds_bytes, g, n0, n, e0, ne, Q, K, V, attn_edge, out);
        DFGNN_TRACE_OUT
        return;
      }
    }
    if (n <= kDenseChunkRows)
      dense_fwd_body<F, WRITE_ATTN, 1, kDenseChunkRows, 1, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_edge, out);
    else if (n <= kDenseWideRows)
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseWideRows, 1, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_edge, out);
    else
      dense_fwd_body<F, WRITE_ATTN, 2, kDenseChunkRows, 2, false, true>(lds, lds_bytes, g, n0, n, e0, ne, 0, g.h, Q, K, V, attn_edge, out);
  }
  DFGNN_TRACE_OUT
#ifdef DFGNN_STAMPS
  if (threadIdx.x == 0 && dfgnn_dense_stamps)
    dfgnn_dense_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 15] = ((unsigned long long)n << 32) | (unsigned)ne;
#endif
}

// GAT 'hyper' forward over the dense ranges: replaces fused_gat_hyper_inference{,_vec4}
// (DFGNN/src/fused_gatconv/fused_gatconv_hyper.cu:5-224) for them.
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gat_dense_fwd_kernel(Csr g, const int *__restrict__ fit,
                                                                      const float *__restrict__ attn_row,
                                                                      const float *__restrict__ attn_col, float slope,
                                                                      const float *__restrict__ X,
                                                                      float *__restrict__ out, int lds_bytes,
                                                                      float *__restrict__ edge_max,
                                                                      float *__restrict__ edge_sum, DenseDrop drop) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if (n <= kDenseChunkRows)
    dense_fwd_body<F, false, 1, kDenseChunkRows, 1, true>(lds, lds_bytes, g, n0, n, e0, ne, blockIdx.y, 1, attn_row, attn_col, X,
                                                          nullptr, out, slope, edge_max, edge_sum, drop);
  else if (n <= kDenseWideRows)
    dense_fwd_body<F, false, 2, kDenseWideRows, 1, true>(lds, lds_bytes, g, n0, n, e0, ne, blockIdx.y, 1, attn_row, attn_col, X,
                                                         nullptr, out, slope, edge_max, edge_sum, drop);
  else
    dense_fwd_body<F, false, 2, kDenseChunkRows, 2, true>(lds, lds_bytes, g, n0, n, e0, ne, blockIdx.y, 1, attn_row, attn_col, X,
                                                          nullptr, out, slope, edge_max, edge_sum, drop);
}

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

template <int FR, int CW, int NBLK, bool GAT = false>
__device__ __forceinline__ void dense_bwd_body(float *lds, const Csr &g, int n0, int n, int e0, int ne, int head,
                                               const float *__restrict__ Q, const float *__restrict__ K,
                                               const float *__restrict__ V, const float *__restrict__ attn_edge,
                                               const float *__restrict__ dO, float *__restrict__ dQ,
                                               float *__restrict__ dK, float *__restrict__ dV,
                                               const GatBwdArgs ga = GatBwdArgs{}) {
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
  const float *attn_h = attn_edge + (size_t)head * g.nnz;
  // GAT: per-node scalars of the range, [SN] each, behind the tile: attn_row, attn_col, edge_max, 1 / edge_sum
  constexpr int SN = NBLK * CW;
  float *smax = T + RBP * TS;            // [8] per-wave maxima of the image being staged, [8] of the dS tile
  float *arl = smax + 2 * kDenseWaves, *acl = arl + SN, *mxl = acl + SN, *ivl = mxl + SN;
  (void)arl, (void)acl, (void)mxl, (void)ivl;

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
  auto image_prefetch = [&](const float *src, int row0, int row_end) {
    if (NBLK > 1) {
      next_src = src;
      next_row0 = row0;
      next_end = row_end;
    } else {
      dense_stage_load<F, CW>(st, src, hf, row0, row_end, fr);
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
      pc[k] = ld32(g.coords + ea0, e);
      if constexpr (!GAT) pa[k] = ld32(attn_h + ea0, e);
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
  constexpr bool kDirectP = !GAT && NBLK == 1;
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
        bc[k] = ld32(g.coords + ea, e);
        if constexpr (!GAT) ba[k] = ld32(attn_h + ea, e);
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

  int ea = e0, eb = (RB < n) ? g.row_ptr[n0 + RB] : e0 + ne;  // edges of the current row block
  edges_prefetch(ea, eb);
  image_prefetch(dOb, 0, min(n, RB));
  tile_open(ea, eb);
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
#pragma unroll
    for (int jc = 0; jc < NBLK; ++jc) {
      const int j0 = jc * CW;
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
      if constexpr (!kDirectP) lds_barrier();  // (the strips' in-place conversions)
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
          for (int r = 0; r < 4; ++r) {
            dS[jc][u][r] = Pr[jc][u][r] * (dS[jc][u][r] - t);
            tmax = fmaxf(tmax, fabsf(dS[jc][u][r]));
          }
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
        ea = eb;
        eb = (i0 + 2 * RB < n) ? g.row_ptr[n0 + i0 + 2 * RB] : e0 + ne;
        image_prefetch(dOb, i0 + RB, min(n, i0 + 2 * RB));
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

// GAT training backward over the dense ranges (no attention dropout): replaces mhspmm_backward_kernel + mhsddmm +
// fused_backward_kernel (DFGNN/src/fused_gatconv/fused_gatconv_kernel.cu:609-865) for them.
template <int F>
__global__ __launch_bounds__(kDenseThreads) void gat_dense_bwd_kernel(
    Csr g, const int *__restrict__ fit, const float *__restrict__ attn_row, const float *__restrict__ attn_col,
    const float *__restrict__ X, const float *__restrict__ dO, float *__restrict__ grad_feat, GatBwdArgs ga) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if (n <= kDenseChunkRows)
    dense_bwd_body<F, kDenseChunkRows, 1, true>(lds, g, n0, n, e0, ne, blockIdx.y, attn_row, attn_col, X, nullptr, dO,
                                                nullptr, nullptr, grad_feat, ga);
  else if (n <= kDenseWideRows)
    dense_bwd_body<F, kDenseWideRows, 1, true>(lds, g, n0, n, e0, ne, blockIdx.y, attn_row, attn_col, X, nullptr, dO,
                                               nullptr, nullptr, grad_feat, ga);
  else
    dense_bwd_body<F, kDenseChunkRows, 2, true>(lds, g, n0, n, e0, ne, blockIdx.y, attn_row, attn_col, X, nullptr, dO,
                                                nullptr, nullptr, grad_feat, ga);
}

template <int F>
__global__ __launch_bounds__(kDenseThreads) void gt_dense_bwd_kernel(
    Csr g, const int *__restrict__ fit, const float *__restrict__ Q, const float *__restrict__ K,
    const float *__restrict__ V, const float *__restrict__ attn_edge, const float *__restrict__ dO,
    float *__restrict__ dQ, float *__restrict__ dK, float *__restrict__ dV, int heads_from, int walk) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  DFGNN_TRACE_IN
  // heads_from < 0: grid (ranges, heads), one workgroup per (range, head).  heads_from >= 0 (multi-head, heads of at
  // most 64 features): a 1-D grid -- the first heads_from ranges (those of more than 128 nodes: they come first in the
  // plan) still get one workgroup per head, every other range ceil(h / walk) workgroups that walk `walk` heads each
  // (dfgnn_dense_heads.hpp).  The 129..160-node ranges stay per head: walked (the body takes NP = 160 for heads of at
  // most 32 features) their workgroups are 8 heads long, 312 of them on 256 CUs, and the kernel waits for the second
  // round -- measured 379 us (8 heads) / 223 us (4 heads) against 387 / 210 us per head.
  // (Workgroups go to the XCDs round-robin by index: the per-head index is range * h + head, so all heads are spread.)
  int range = blockIdx.x, head = blockIdx.y, nwalk = 0;
  if (heads_from >= 0) {
    const int w = blockIdx.x;
    if (w < heads_from * g.h) {
      range = w / g.h;
      head = w - range * g.h;
    } else {
      const int chunks = (g.h + walk - 1) / walk, k = w - heads_from * g.h;
      range = heads_from + k / chunks;
      head = (k % chunks) * walk;
      nwalk = min(walk, g.h - head);
    }
  }
  const int n0 = fit[2 * range], n1 = fit[2 * range + 1] & kPlanRangeMask;
  const int n = n1 - n0, e0 = g.row_ptr[n0], ne = g.row_ptr[n1] - e0;
  if constexpr (F <= 64) {
    if (nwalk > 0) {  // (n <= 128 by construction)
      dense_bwd_heads_body<F>(lds, g, n0, n, e0, ne, head, nwalk, Q, K, V, attn_edge, dO, dQ, dK, dV);
      DFGNN_TRACE_OUT
      return;
    }
  }
#ifdef DFGNN_BWD_VARIANT  // diagnostic builds: one geometry only (register use / ISA of a single body)
  constexpr int only = DFGNN_BWD_VARIANT;
#else
  constexpr int only = -1;
#endif
  if ((only < 0 && n <= kDenseChunkRows) || only == 0)
#ifdef DFGNN_RING128  // A/B builds: half-width images through the prefetch ring for the <= 128-node ranges as well
    dense_bwd_wide_body<F, kDenseChunkRows, DFGNN_RING128, DFGNN_FW128>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#else
    dense_bwd_body<F, kDenseChunkRows, 1>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#endif
  else if ((only < 0 && n <= kDenseWideRows) || only == 1)
#ifdef DFGNN_OLD_WIDE_BWD  // A/B builds only: two row blocks of <= 80 rows, full-width images
    dense_bwd_body<F, kDenseWideRows, 1>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#else
    dense_bwd_wide_body<F, kDenseWideRows, DFGNN_RING160>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
#endif
  else
    dense_bwd_body<F, kDenseChunkRows, 2>(lds, g, n0, n, e0, ne, head, Q, K, V, attn_edge, dO, dQ, dK, dV);
  DFGNN_TRACE_OUT
#ifdef DFGNN_STAMPS
  if (threadIdx.x == 0 && dfgnn_dense_stamps)
    dfgnn_dense_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + 15] = ((unsigned long long)n << 32) | (unsigned)ne;
#endif
}

// =====================================================================================================================
// launchers: the first p.num_dense entries of the plan's fit list
// =====================================================================================================================
bool dense_enabled() {
  static const bool on = [] { const char *e = getenv("DFGNN_DENSE"); return !e || atoi(e) != 0; }();
  return on;
}

template <class Fn>
static int dispatch_dense(int f, Fn &&fn) {
  if (f == 8) return fn(std::integral_constant<int, 8>{});    // f = 8 / 16: zero-padded onto the 32-wide layout
  if (f == 16) return fn(std::integral_constant<int, 16>{});
  if (f == 32) return fn(std::integral_constant<int, 32>{});
  if (f == 64) return fn(std::integral_constant<int, 64>{});
  if (f == 128) return fn(std::integral_constant<int, 128>{});
  return kErrUnsupported;
}

// Heads a workgroup of the multi-head backward walks (ranges of <= 128 nodes): all of them (8 heads of 16: 393 us with
// 8, 397 with 4, 408 with 2 heads per workgroup, 439 per head).  DFGNN_HEADS_WALK in the environment (diagnostic switch,
// read once) overrides it; 0 = one workgroup per (range, head) throughout.
static int heads_walk() {
  static const int w = [] { const char *e = getenv("DFGNN_HEADS_WALK"); return e ? max(0, atoi(e)) : 64; }();
  return w;
}

// DFGNN_LEAN=0 in the environment (diagnostic switch, read once): every dense range on the 512-thread forward
static bool lean_enabled() {
  static const bool on = [] { const char *e = getenv("DFGNN_LEAN"); return !e || atoi(e) != 0; }();
  return on;
}

static int launch_gt_dense_fwd_single(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                                      float *attn_edge, float *out, hipStream_t s) {
  Csr g = g_in;
  g.coords = p.coords();
  const dim3 grid(p.num_dense, 1);
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (attn_edge) {
      if (int rc = set_max_lds(gt_dense_fwd_kernel<F, true>)) return rc;
      gt_dense_fwd_kernel<F, true><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_edge, out, kLdsBytes);
    } else {
      if (int rc = set_max_lds(gt_dense_fwd_kernel<F, false>)) return rc;
      gt_dense_fwd_kernel<F, false><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, nullptr, out, kLdsBytes);
    }
    return launch_status();
  });
}

// The ranges of more than 128 nodes need the 512-thread kernel (one workgroup per CU); a batch without any takes the
// 256-thread kernel, two workgroups per CU (dfgnn_dense_lean.hpp: ~12 % faster on such batches).  A MIXED batch stays on the
// 512-thread kernel as a whole: run as two kernels the classes serialise -- one after the other on the caller's stream the
// second waits for the first one's tail (forward 108 -> 130 us on the headline batch), and forked onto a side stream of the
// library's own (fork / join events) the two grids still ran back to back on this runtime (127 us).
int launch_gt_dense_fwd(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                        float *attn_edge, float *out, hipStream_t s) {
  if (p.num_dense == 0) return 0;
  const bool lean = (g_in.f == 64 || g_in.f == 128) && g_in.h == 1 && p.num_dense_wide == 0 && lean_enabled();
  if (!lean) return launch_gt_dense_fwd_single(g_in, p, Q, K, V, attn_edge, out, s);
  Csr g = g_in;
  g.coords = p.coords();
  const dim3 grid(p.num_dense, 1);
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if constexpr (F == 64 || F == 128) {
      if (attn_edge) {
        if (int rc = set_max_lds(gt_dense_fwd_lean_kernel<F, true>)) return rc;
        gt_dense_fwd_lean_kernel<F, true><<<grid, kLeanThreads, kLeanLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_edge, out);
      } else {
        if (int rc = set_max_lds(gt_dense_fwd_lean_kernel<F, false>)) return rc;
        gt_dense_fwd_lean_kernel<F, false><<<grid, kLeanThreads, kLeanLdsBytes, s>>>(g, p.fit(), Q, K, V, nullptr, out);
      }
    }
    return launch_status();
  });
}

int launch_gat_dense_fwd(const Csr &g_in, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, float *out, hipStream_t s, float *edge_max, float *edge_sum,
                         const float *edge_mask, float attn_drop) {
  Csr g = g_in;
  g.coords = p.coords();
  if (p.num_dense == 0) return 0;
  const dim3 grid(p.num_dense, g.h);
  DenseDrop drop;
  if (edge_mask) drop = DenseDrop{edge_mask, attn_drop, 1.f / (1.f - attn_drop)};
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds(gat_dense_fwd_kernel<F>)) return rc;
    gat_dense_fwd_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), attn_row, attn_col, slope, X, out, kLdsBytes,
                                                                   edge_max, edge_sum, drop);
    return launch_status();
  });
}

int launch_gat_dense_bwd(const Csr &g_in, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, const float *edge_max, const float *edge_sum, const float *grad_out,
                         float *grad_feat, float *grad_row, float *grad_col, hipStream_t s, const float *edge_mask,
                         float attn_drop) {
  Csr g = g_in;
  g.coords = p.coords();
  if (p.num_dense == 0) return 0;
  const dim3 grid(p.num_dense, g.h);
  GatBwdArgs ga{edge_max, edge_sum, slope, grad_row, grad_col, DenseDrop{}};
  if (edge_mask) ga.drop = DenseDrop{edge_mask, attn_drop, 1.f / (1.f - attn_drop)};
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds(gat_dense_bwd_kernel<F>)) return rc;
    gat_dense_bwd_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), attn_row, attn_col, X, grad_out, grad_feat, ga);
    return launch_status();
  });
}

int launch_gt_dense_bwd(const Csr &g_in, const Plan &p, const float *Q, const float *K, const float *V,
                        const float *attn_edge, const float *grad_out, float *dQ, float *dK, float *dV,
                        hipStream_t s) {
  Csr g = g_in;
  g.coords = p.coords();
  if (p.num_dense == 0) return 0;
  // multi-head, heads of at most 64 features: one workgroup per range of <= 128 nodes walks the heads (see the kernel)
  const int walk = (g.h > 1 && g.f <= 64) ? min(heads_walk(), g.h) : 0;
  const int heads_from = walk ? p.num_dense_wide : -1;
  const int chunks = walk ? (g.h + walk - 1) / walk : 0;
  const dim3 grid = walk ? dim3(p.num_dense_wide * g.h + (p.num_dense - p.num_dense_wide) * chunks, 1) : dim3(p.num_dense, g.h);
  return dispatch_dense(g.f, [&](auto fc) {
    constexpr int F = decltype(fc)::value;
    if (int rc = set_max_lds(gt_dense_bwd_kernel<F>)) return rc;
    gt_dense_bwd_kernel<F><<<grid, kDenseThreads, kLdsBytes, s>>>(g, p.fit(), Q, K, V, attn_edge, grad_out, dQ, dK, dV,
                                                                  heads_from, walk);
    return launch_status();
  });
}

}  // namespace dfgnn

#ifdef DFGNN_STAMPS
extern "C" int dfgnn_debug_set_dense_stamps(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_dense_stamps), &p, sizeof(p));
}
extern "C" int dfgnn_debug_set_wg_trace(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_wg_trace), &p, sizeof(p));
}
#endif
