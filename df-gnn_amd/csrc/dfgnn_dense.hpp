// dfgnn_dense.hpp -- building blocks of the matrix-core ("dense") GT kernels (gt_dense.hip).
//
// A closed node range of a batched graph is a small square attention problem: n <= 255 nodes, and for the
// dense graphs the headline workload is made of (PATTERN: ~119 nodes, 44 % of all pairs are edges) most of the
// n x n logits are real.  Computing ALL of them on the matrix cores and masking costs ~0.1 cycle per node pair
// per CU; walking the edges on the VALU costs ~7 cycles per edge.  The block plan (plan.hip) therefore marks the
// ranges that have at least one edge per 32 node pairs, at most 255 nodes and no duplicate edges as "dense"; with
// unit edge values they are served by the kernels of gt_dense.hip, everything else by the edge-walking kernels.
//
// Numerics (fp32-equivalent): every operand is multiplied by a power of two that brings the largest magnitude of its
// image / strip / tile into [2^14, 2^15) and split into two fp16 halves, x s = hi + lo with hi = rn16(x s),
// lo = rn16(x s - hi): |x s - hi - lo| <= 2^-24 |x s| for every element within 2^-17 of the maximum (below that the
// error is the fp16 subnormal spacing, 2^-39 of the maximum).  A product X Y is accumulated in fp32 as
// Xhi Yhi + Xhi Ylo + Xlo Yhi on v_mfma_f32_16x16x32_f16 (the dropped lo lo term is <= 2^-24 of the product) and the
// result is multiplied by the inverse powers of two, which is exact: the error per product is ~3 x 2^-24, that of an
// fp32 FMA chain of the same length.  (The first version used bf16 halves: 16 significant bits, no scaling.)
//
// Layouts.  A feature matrix is staged 128 rows at a time as a row-major fp16 "image" (hi and lo copies, rows
// skewed by 16 elements so that both the 16-byte row reads and the transposed reads are bank-conflict free).
// For v_mfma_f32_16x16x32_f16, lane l = (mi = l & 15, mq = l >> 4) holds A[row mi][k = 8 mq + t] and
// B[k = 8 mq + t][col mi] in element t, and D[row 4 mq + r][col mi] in register r.
//   * rows of an image as the A operand: one 16-byte read per k-step  ->  D^T tiles (rows = image rows, cols = the
//     16 rows of the register operand), i.e. a wave that owns 16 rows i ("strip") of S = Q K^T gets S[i][j] for its
//     lane's i = mi and j = 16 tile + 4 mq + r: a row of S lives on the 4 lanes {mi, mi+16, mi+32, mi+48}.
//   * those accumulators ARE the B operand of the next product (O^T = V^T P^T) with the k-slots permuted: element t of
//     lane (mi, mq) <-> column 32 jb + 4 mq + t (t < 4) or 32 jb + 16 + 4 mq + t - 4; the other operand (V^T) is read
//     with the same permutation through ds_read_b64_tr_b16 from the row-major V image.
//   * the O^T accumulator holds 4 consecutive features of one output row per lane -> float4 stores.
#pragma once
#include "dfgnn_block.hpp"

namespace dfgnn {

typedef _Float16 h16;                                         // operand halves
typedef __attribute__((ext_vector_type(8))) _Float16 hx8;
typedef __attribute__((ext_vector_type(4))) _Float16 hx4;
typedef __fp16 fp16x4_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));  // the transposed-read builtin's type
typedef __attribute__((address_space(3))) fp16x4_raw *lds_hx4_ptr;

// ---- debug build: bounds-checked LDS addressing ---------------------------------------------------------------------
// `make ldscheck` compiles the matrix-core translation units with -DDFGNN_LDS_CHECK into libdfgnn_ldscheck.so.  Every
// LDS address the helpers of this file form (image stores, row / transposed operand reads, the ld32 / st32 accessors when
// they are handed an LDS base, the workgroup-maximum slots) is then compared with the workgroup's LDS allocation -- the
// kernel's static size plus the launch's dynamic size (hidden_dynamic_lds_size, implicit kernel argument 30 of code
// object v5) -- before it is used.  A violation is COUNTED, not trapped (a trapping kernel can take the GPU down for
// everyone on the host; the hardware itself drops out-of-range LDS accesses): the first one's source line, byte offset and
// limit are kept and dfgnn_debug_lds_report() (capi.hip, that build only) hands them to the host, summed over the
// translation units.  tests/test_gpu_ldscheck.py runs the parity workloads through that library.
#if defined(DFGNN_LDS_CHECK)
__attribute__((used)) static __device__ unsigned g_lds_report[4];  // violations, line of the first, its end offset, the limit
__device__ __forceinline__ void lds_check(const void *p, unsigned bytes, int line) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (!__builtin_amdgcn_is_shared(p)) return;                      // a global base handed to ld32 / st32
  const unsigned end = (unsigned)(size_t)p + bytes;               // low half of a generic LDS address = byte offset
  const unsigned lim = __builtin_amdgcn_groupstaticsize() + ((const unsigned *)__builtin_amdgcn_implicitarg_ptr())[30];
  if (end > lim && atomicAdd(&g_lds_report[0], 1u) == 0u) {
    g_lds_report[1] = (unsigned)line;
    g_lds_report[2] = end;
    g_lds_report[3] = lim;
  }
#endif
}
#define DFGNN_LDS_AT(p, bytes) ::dfgnn::lds_check((p), (bytes), __LINE__)
void lds_report_register(const void *symbol);  // gt_dense.hip: the list dfgnn_debug_lds_report() walks
namespace {
struct LdsReportReg {  // one per translation unit that includes this header: each owns a copy of g_lds_report
  LdsReportReg() { lds_report_register(HIP_SYMBOL(g_lds_report)); }
} g_lds_report_reg;
}  // namespace
#else
#define DFGNN_LDS_AT(p, bytes) ((void)0)
#endif

// Power-of-two scale of an operand whose largest magnitude is amax: s amax lies in [2^14, 2^15), inv = 1 / s.
// (amax = 0 or subnormal: the largest normal scale; inf / nan inputs give inf / nan outputs, as in fp32.)
struct Pow2Scale {
  float s, inv;
};
__device__ __forceinline__ Pow2Scale pow2_scale(float amax) {  // amax must be wave-uniform: the result lives in SGPRs
  const int b = min(max((__builtin_amdgcn_readfirstlane(__float_as_int(amax)) >> 23) & 0xFF, 15), 254);  // biased exponent
  return Pow2Scale{__int_as_float((268 - b) << 23), __int_as_float((b - 14) << 23)};
}
constexpr float kUnitScale = 16384.f, kUnitScaleInv = 1.f / 16384.f;  // for operands known to lie in [0, 1]

// x s -> fp16 hi / lo halves (v_pk_mul_f32, v_cvt_pk_f16_f32, v_cvt_f32_f16, v_pk_fma_f32, v_cvt_pk_f16_f32)
__device__ __forceinline__ void split_hx8(const float4 &a, const float4 &b, float s, hx8 &hi, hx8 &lo) {
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const h16 h = (h16)(x[j] * s);
    hi[j] = h;
    lo[j] = (h16)fmaf(x[j], s, -(float)h);
  }
}
__device__ __forceinline__ float absmax8(const float4 &a, const float4 &b) {
  return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
               fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
}

constexpr int kDenseThreads = 512;                    // 8 waves: up to 256 VGPRs each, one workgroup per CU (LDS)
constexpr int kDenseWaves = kDenseThreads / kWave;
constexpr int kDenseChunkRows = 128;                  // rows of a feature matrix resident at a time ...
constexpr int kDenseWideRows = 160;                   // ... or 160, which lets ranges of 129-160 nodes do without a second chunk
constexpr int kDensePre = 16;                         // edges per thread fetched ahead of the scatter loops

template <int F>
struct DenseCfg {
  static constexpr int RS = F + 16;  // fp16 elements per image row
  static constexpr int KT = F / 32;  // MFMA k-steps across the feature dimension
  static constexpr int FT = F / 16;  // 16-feature tiles of the output
};

// base[idx] with a 32-bit byte offset: lets the compiler keep `base` in SGPRs and the offset in ONE VGPR
// (global_load ... v_off, s[base]) instead of materialising a 64-bit address per access.
template <class T>
__device__ __forceinline__ const T &ld32(const T *base, unsigned idx) {
  DFGNN_LDS_AT(reinterpret_cast<const char *>(base) + (size_t)(idx * (unsigned)sizeof(T)), (unsigned)sizeof(T));
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(idx * (unsigned)sizeof(T)));
}
__device__ __forceinline__ float4 ld32_f4(const float *base, unsigned idx) {  // 4 floats starting at base[idx]
  DFGNN_LDS_AT(reinterpret_cast<const char *>(base) + (size_t)(idx * 4u), 16u);
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(base) + (size_t)(idx * 4u));
}
__device__ __forceinline__ void st32_f4(float *base, unsigned idx, const float4 &v) {
  DFGNN_LDS_AT(reinterpret_cast<char *>(base) + (size_t)(idx * 4u), 16u);
  *reinterpret_cast<float4 *>(reinterpret_cast<char *>(base) + (size_t)(idx * 4u)) = v;
}
// Loads of data these kernels read ONCE and nothing of the pair reads again (dO, the attention values and edge coordinates
// of a backward): non-temporal, for the same reason as the stores below.  -DDFGNN_NT_LOADS=0: plain loads (A/B).
#ifndef DFGNN_NT_LOADS
#define DFGNN_NT_LOADS 1
#endif
template <class T>
__device__ __forceinline__ T ld32_once(const T *base, unsigned idx) {
  const T *p = reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(idx * (unsigned)sizeof(T)));
#if DFGNN_NT_LOADS
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ float4 ld32_f4_once(const float *base, unsigned idx) {
#if DFGNN_NT_LOADS
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v x = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(reinterpret_cast<const char *>(base) + (size_t)(idx * 4u)));
  return make_float4(x[0], x[1], x[2], x[3]);
#else
  return ld32_f4(base, idx);
#endif
}
// The same for an OUTPUT row in global memory (out, dQ, dK, dV: written once, not read again by these kernels), as a
// NON-TEMPORAL store: the outputs then do not push the inputs out of L2 / the Infinity Cache.  A step moves 270 MB (forward)
// and 490 MB (backward) past a 256 MB cache; with 62 / 185 MB of that streamed, the backward finds more of what the forward
// read and the next forward more of what the backward read: step 222.5 -> 211.7 us in the kernel trace (forward 88.5 ->
// 84.7, backward 134.0 -> 127.0).  -DDFGNN_NT_STORES=0 builds with plain stores (A/B).
#ifndef DFGNN_NT_STORES
#define DFGNN_NT_STORES 1
#endif
__device__ __forceinline__ void st32_f4_out(float *base, unsigned idx, const float4 &v) {
#if DFGNN_NT_STORES
  typedef float f4v __attribute__((ext_vector_type(4)));
  f4v x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<f4v *>(reinterpret_cast<char *>(base) + (size_t)(idx * 4u)));
#else
  st32_f4(base, idx, v);
#endif
}

// threadIdx.x / the MFMA lane coordinates behind an optimisation barrier, taken afresh by every phase: index
// arithmetic derived from them is then recomputed where it is used instead of being hoisted out of the phase loops
// and kept (and spilled) for the whole kernel.
__device__ __forceinline__ int opaque_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
struct LaneIds {
  int mi, mq, tq, tp;  // lane & 15, lane >> 4, and the (row, 4-column group) a lane addresses in a transposed read
};
__device__ __forceinline__ LaneIds lane_ids() {
  int l = threadIdx.x & (kWave - 1);
  asm volatile("" : "+v"(l));
  return LaneIds{l & 15, l >> 4, (l & 15) >> 2, l & 3};
}

// ---- feature images -----------------------------------------------------------------------------------------------
// ROWS rows x F features: global -> registers (issued one phase ahead of its use) -> fp16 hi / lo images in LDS.
template <int F, int ROWS>
struct DenseStageRegs {
  static constexpr int PER = (ROWS * (F / 8) + kDenseThreads - 1) / kDenseThreads;  // 8-float pieces per thread
  float4 a[PER], b[PER];
  int row0, row_end;  // the image holds rows [row0, row0 + ROWS) of the matrix; rows at or past row_end are zero
};

// rows [row0, row0 + ROWS) of the matrix whose row 0 is `src` (global row stride hf floats); rows at or past row_end
// (> row0) read as zero.  The loads are unconditional (addresses clamped to the last valid row, the zeroing happens
// when the image is stored): a load under a branch is waited for right behind the branch, one round trip per piece.
// fr < F (a narrower matrix run on the F-wide kernels, e.g. f = 16 heads on the 32-wide instance): the pieces past fr are
// read from the row's first piece instead (a valid address) and zeroed with the padding rows.
// once (a literal at every call site, so that it folds): the matrix is read once and not again -- non-temporal loads.
template <int F, int ROWS>
__device__ __forceinline__ void dense_stage_load(DenseStageRegs<F, ROWS> &r, const float *__restrict__ src, size_t hf,
                                                 int row0, int row_end, int fr = F, bool once = false) {
  constexpr int C8 = F / 8;
  const int tid = opaque_tid();
  r.row0 = row0;
  r.row_end = row_end;
#pragma unroll
  for (int k = 0; k < DenseStageRegs<F, ROWS>::PER; ++k) {
    const int idx = tid + k * kDenseThreads;
    const int row = min(idx / C8, ROWS - 1), c8 = idx % C8;
    const unsigned off = (unsigned)min(row0 + row, row_end - 1) * (unsigned)hf + ((fr >= F || 8 * c8 < fr) ? 8u * c8 : 0u);
    r.a[k] = once ? ld32_f4_once(src, off) : ld32_f4(src, off);
    r.b[k] = once ? ld32_f4_once(src, off + 4) : ld32_f4(src, off + 4);
  }
}

// Largest magnitude this thread holds of the image that is about to be stored.  (Clamped loads repeat valid rows and
// pieces, so nothing needs masking.)
template <int F, int ROWS>
__device__ __forceinline__ float dense_stage_absmax(const DenseStageRegs<F, ROWS> &r) {
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < DenseStageRegs<F, ROWS>::PER; ++k) m = fmaxf(m, absmax8(r.a[k], r.b[k]));
  return m;
}

// Workgroup-wide maximum through an 8-entry LDS array: every wave posts its own maximum BEFORE a workgroup barrier the
// caller already has, every thread reads all of them after it.  `slot` must not be posted to again before another
// barrier has passed (every caller has one: the barrier that publishes what was converted with the result).
__device__ __forceinline__ void wg_max_post(float *slot, float v) {
  v = wave_max(v);
  DFGNN_LDS_AT(slot + threadIdx.x / kWave, 4u);
  if ((threadIdx.x & (kWave - 1)) == 0) slot[threadIdx.x / kWave] = v;
}
__device__ __forceinline__ float wg_max_read(const float *slot) {
  DFGNN_LDS_AT(slot, 32u);
  const float4 a = *reinterpret_cast<const float4 *>(slot), b = *reinterpret_cast<const float4 *>(slot + 4);
  return fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
}

// The trailing memory clobber keeps the loads of the next prefetch (usually into these same registers) from being
// interleaved with the stores piece by piece: memory returns in order, so each piece would then wait for the loads
// that were just issued.  (Pinning the conversion here with an empty asm, so that the scheduler cannot hoist it
// and the wait for the data to the previous barrier, measured 2 % slower.)
// fr: real feature count (<= F, a multiple of 8): the image columns at or past it are stored as zeros.
// scale: the image's power-of-two scale (pow2_scale of its largest magnitude).
template <int F, int ROWS>
__device__ __forceinline__ void dense_stage_store(DenseStageRegs<F, ROWS> &r, h16 *hi, h16 *lo, float scale, int fr = F) {
  constexpr int C8 = F / 8, RS = DenseCfg<F>::RS;
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
      DFGNN_LDS_AT(hi + row * RS + 8 * c8, 16u);
      DFGNN_LDS_AT(lo + row * RS + 8 * c8, 16u);
      *reinterpret_cast<hx8 *>(hi + row * RS + 8 * c8) = h;
      *reinterpret_cast<hx8 *>(lo + row * RS + 8 * c8) = l;
    }
  }
  asm volatile("" ::: "memory");
}

// D^T tile u (image rows 16 u .. 16 u + 15) against a register row operand: 3 * F/32 MFMAs
template <int F>
__device__ __forceinline__ void dense_rows_frag(hx8 (&ah)[F / 32], hx8 (&al)[F / 32], const h16 *ihi,
                                                const h16 *ilo, int u, const LaneIds &L) {
  constexpr int RS = DenseCfg<F>::RS;
  const int off = (16 * u + L.mi) * RS + 8 * L.mq;
#pragma unroll
  for (int t = 0; t < F / 32; ++t) {
    DFGNN_LDS_AT(ihi + off + 32 * t, 16u);
    DFGNN_LDS_AT(ilo + off + 32 * t, 16u);
    ah[t] = *reinterpret_cast<const hx8 *>(ihi + off + 32 * t);
    al[t] = *reinterpret_cast<const hx8 *>(ilo + off + 32 * t);
  }
}
template <int F>
__device__ __forceinline__ f32x4 dense_rows_mma_frag(const hx8 (&ah)[F / 32], const hx8 (&al)[F / 32],
                                                     const hx8 (&xh)[F / 32], const hx8 (&xl)[F / 32]) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < F / 32; ++t) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], xh[t], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t], xl[t], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t], xh[t], acc, 0, 0, 0);
  }
  return acc;
}
template <int F>
__device__ __forceinline__ f32x4 dense_rows_mma(const h16 *ihi, const h16 *ilo, int u, const hx8 (&xh)[F / 32],
                                                const hx8 (&xl)[F / 32], const LaneIds &L) {
  hx8 ah[F / 32], al[F / 32];
  dense_rows_frag<F>(ah, al, ihi, ilo, u, L);
  return dense_rows_mma_frag<F>(ah, al, xh, xl);
}

// D^T tiles 0 .. NTILES-1 of the resident image (those with 16 u < limit; the others are zero) against a register
// row operand, the image fragments of tile u + 1 being fetched while tile u is multiplied
template <int F, int NTILES>
__device__ __forceinline__ void dense_rows_mma_strip(f32x4 (&out)[NTILES], const h16 *ihi, const h16 *ilo, int limit,
                                                     const hx8 (&xh)[F / 32], const hx8 (&xl)[F / 32],
                                                     const LaneIds &L) {
  hx8 ah[2][F / 32], al[2][F / 32];
  dense_rows_frag<F>(ah[0], al[0], ihi, ilo, 0, L);
#pragma unroll
  for (int u = 0; u < NTILES; ++u) {
    if (u + 1 < NTILES) dense_rows_frag<F>(ah[(u + 1) & 1], al[(u + 1) & 1], ihi, ilo, u + 1, L);
    out[u] = (16 * u < limit) ? dense_rows_mma_frag<F>(ah[u & 1], al[u & 1], xh, xl) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

// Two transposed 4-row reads -> one 8-element operand fragment (rows r .. r+3 and r + second .. of a column).
__device__ __forceinline__ hx8 dense_tr_pair(const h16 *p, int second_offset) {
  DFGNN_LDS_AT(p, 8u);
  DFGNN_LDS_AT(p + second_offset, 8u);
  const hx4 a = __builtin_bit_cast(hx4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hx4_ptr)(p)));
  const hx4 b = __builtin_bit_cast(hx4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hx4_ptr)(p + second_offset)));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 8 fp32 values x scale -> fp16 hi / lo operand fragments
__device__ __forceinline__ void dense_split8(const f32x4 &x0, const f32x4 &x1, float scale, hx8 &h, hx8 &l) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float x = (t < 4) ? x0[t] : x1[t - 4];
    const h16 hh = (h16)(x * scale);
    h[t] = hh;
    l[t] = (h16)fmaf(x, scale, -(float)hh);
  }
}

// acc[ft] += X^T Y for one 32-deep k-block: X^T fragments come from the image through transposed reads (`xoff` =
// this lane's element offset of feature tile 0, `second` = element offset between its two 4-row reads), Y is given
// as operand fragments.  The X fragments of GMAX (four or eight) feature tiles are fetched together and the products
// are issued as three sweeps over that many independent accumulators, so neither the LDS latency nor the MFMA result latency
// serialises the chain.
template <int F, int GMAX = 4>
__device__ __forceinline__ void dense_kblock_mma(f32x4 (&acc)[F / 16], const h16 *ihi, const h16 *ilo, int xoff,
                                                 int second, const hx8 &yh, const hx8 &yl) {
  constexpr int FT = F / 16, G = FT < GMAX ? FT : GMAX;
#pragma unroll
  for (int f0 = 0; f0 < FT; f0 += G) {
    hx8 xh[G], xl[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {
      xh[k] = dense_tr_pair(ihi + xoff + 16 * (f0 + k), second);
      xl[k] = dense_tr_pair(ilo + xoff + 16 * (f0 + k), second);
    }
#pragma unroll
    for (int k = 0; k < G; ++k) acc[f0 + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yh, acc[f0 + k], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < G; ++k) acc[f0 + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[k], yh, acc[f0 + k], 0, 0, 0);
#pragma unroll
    for (int k = 0; k < G; ++k) acc[f0 + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yl, acc[f0 + k], 0, 0, 0);
  }
}

// The same k-block for TWO operand fragments Y0, Y1 (two 16-column strips of the other matrix) and NFT feature tiles
// starting at tile ft0: the X fragments are fetched once and used twice.  A wave that owns 2 strips x F/32 tiles instead of
// 1 strip x F/16 tiles reads a third less from LDS per MFMA (operand fetches ~ perimeter of the register block), and
// these products are bound by the LDS fragment traffic of the hi / lo images, not by the matrix pipe.
template <int NFT>
__device__ __forceinline__ void dense_kblock_mma2(f32x4 (&acc0)[NFT], f32x4 (&acc1)[NFT], const h16 *ihi, const h16 *ilo,
                                                  int xoff, int second, const hx8 &yh0, const hx8 &yl0, const hx8 &yh1,
                                                  const hx8 &yl1) {
  hx8 xh[NFT], xl[NFT];
#pragma unroll
  for (int k = 0; k < NFT; ++k) {
    xh[k] = dense_tr_pair(ihi + xoff + 16 * k, second);
    xl[k] = dense_tr_pair(ilo + xoff + 16 * k, second);
  }
#pragma unroll
  for (int k = 0; k < NFT; ++k) acc0[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yh0, acc0[k], 0, 0, 0);
#pragma unroll
  for (int k = 0; k < NFT; ++k) acc1[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yh1, acc1[k], 0, 0, 0);
#pragma unroll
  for (int k = 0; k < NFT; ++k) acc0[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[k], yh0, acc0[k], 0, 0, 0);
#pragma unroll
  for (int k = 0; k < NFT; ++k) acc1[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[k], yh1, acc1[k], 0, 0, 0);
#pragma unroll
  for (int k = 0; k < NFT; ++k) acc0[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yl0, acc0[k], 0, 0, 0);
#pragma unroll
  for (int k = 0; k < NFT; ++k) acc1[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[k], yl1, acc1[k], 0, 0, 0);
}

// acc[ft] += Image^T (features 16 ft .., image rows 32 jb ..) . Y, Y given as the accumulator pair (y0, y1) of a
// D^T strip (permuted k order, see the header): the "P V" product of the forward.  yscale: power-of-two scale of Y.
template <int F, int GMAX = 4>
__device__ __forceinline__ void dense_cols_mma(f32x4 (&acc)[F / 16], const h16 *ihi, const h16 *ilo, int jb,
                                               const f32x4 &y0, const f32x4 &y1, float yscale, const LaneIds &L) {
  constexpr int RS = DenseCfg<F>::RS;
  hx8 yh, yl;
  dense_split8(y0, y1, yscale, yh, yl);
  dense_kblock_mma<F, GMAX>(acc, ihi, ilo, (32 * jb + 4 * L.mq + L.tq) * RS + 4 * L.tp, 16 * RS, yh, yl);
}

// accumulator tiles -> a global row: lane (mi, mq), register r of tile ft = feature 16 ft + 4 mq + r.
// GUARD (padded widths only): feat0 = the feature of this lane's first value (4 mq, plus the tile offset of a
// single-tile call), fr = the real feature count: nothing is stored (or read back) at or past it.  Unguarded, the
// stores and the read-backs stay free of branches (a load under a branch is waited for right behind it).
template <int NFT, bool GUARD = false>
__device__ __forceinline__ void dense_store_acc(const f32x4 (&acc)[NFT], float scale, float *__restrict__ base,
                                                unsigned off, bool accumulate, int feat0 = 0, int fr = 1 << 30) {
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft) {
    if constexpr (GUARD)
      if (feat0 + 16 * ft >= fr) continue;
    float4 o = make_float4(acc[ft][0] * scale, acc[ft][1] * scale, acc[ft][2] * scale, acc[ft][3] * scale);
    if (accumulate) {
      const float4 old = ld32_f4(base, off + 16 * ft);
      o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
    }
    st32_f4_out(base, off + 16 * ft, o);
  }
}

// accumulator tiles of one 16-row strip -> global rows, in WHOLE 128-byte lines: a tile gives a row 64 contiguous bytes
// (4 lanes x float4), so storing tile by tile makes every store instruction write 16 half lines -- and a stream of
// half-line writes costs the whole kernel ~15 % of the HBM rate it can reach (tools/diag/stream_probe.hip).  Here two
// neighbouring tiles are stored together: lanes with mi < 8 take tile 2k of their own row, lanes with mi >= 8 take tile
// 2k + 1 of row mi - 8 (one row_ror:8 DPP move per register), so the first instruction writes rows 0..7 of the strip
// and the second rows 8..15, 128 contiguous bytes per row.  Must be called by all lanes of the wave; `row` = this
// lane's row of the matrix (strip * 16 + mi), rows >= n are not stored.  NFT even; feature f of tile t = 16 t + 4 mq.
constexpr int kDppRowRor8 = 0x128;
template <int NFT>
__device__ __forceinline__ void dense_store_rows(const f32x4 (&acc)[NFT], float scale, float *__restrict__ base, unsigned hf,
                                                 int row, int n, const LaneIds &L) {
  static_assert(NFT % 2 == 0, "tiles are stored in pairs");
  const bool low = L.mi < 8;
  const int prow = low ? row + 8 : row - 8;  // the partner lane's row
  const int r1 = low ? row : prow, r2 = low ? prow : row;
  const unsigned o1 = (unsigned)r1 * hf + (low ? 0u : 16u) + 4u * L.mq, o2 = (unsigned)r2 * hf + (low ? 16u : 0u) + 4u * L.mq;
#pragma unroll
  for (int k = 0; k < NFT / 2; ++k) {
    float4 a, b;
    a.x = acc[2 * k][0] * scale; a.y = acc[2 * k][1] * scale; a.z = acc[2 * k][2] * scale; a.w = acc[2 * k][3] * scale;
    b.x = dpp_perm<kDppRowRor8>(acc[2 * k + 1][0] * scale);
    b.y = dpp_perm<kDppRowRor8>(acc[2 * k + 1][1] * scale);
    b.z = dpp_perm<kDppRowRor8>(acc[2 * k + 1][2] * scale);
    b.w = dpp_perm<kDppRowRor8>(acc[2 * k + 1][3] * scale);
    const float4 v1 = low ? a : b, v2 = low ? b : a;
    if (r1 < n) st32_f4_out(base, o1 + 32 * k, v1);
    if (r2 < n) st32_f4_out(base, o2 + 32 * k, v2);
  }
}

// All-reduce over the four lanes {mi, mi + 16, mi + 32, mi + 48} that hold one row of a D^T strip: two permlane swaps
// (VALU rate) instead of two ds_bpermute round trips through the LDS crossbar.  swap32 of two copies leaves the lower
// half's values in both halves of one and the upper half's in the other; swap16 does the same for the 16-lane rows.
__device__ __forceinline__ float xor16_32_max(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  a = fmaxf(a, b);
  b = a;
  permlane16_swap(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xor16_32_sum(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  a += b;
  b = a;
  permlane16_swap(a, b);
  return a + b;
}

}  // namespace dfgnn
