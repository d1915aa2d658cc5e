// dfgnn_device.hpp -- wave64 building blocks shared by every kernel of libdfgnn (gfx950 only).
//
// Vocabulary
//   group  : G consecutive lanes of a wave that together hold one feature row ([f] floats) as
//            NCH chunks of VEC floats per lane.  A wave holds EPW = 64 / G rows at once, so a
//            wave touches EPW edges per gather instruction (f = 128 -> G = 32 lanes x float4,
//            two 512-byte rows per wave-instruction, fully coalesced).
//   Frag   : the per-lane slice of one feature row.
//
// This replaces the reference's 32-lane-warp idioms (DFGNN/src/util/computeUtil.h:
// WARP_SIZE=32, __shfl_*_sync(...,32)); nothing here is a translation of that header.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dfgnn {

constexpr int kWave = 64;
constexpr int kBlock = 256;               // 4 waves / workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

__device__ __forceinline__ float fast_exp(float x) {
  // v_exp_f32 (2^x); relative error ~1e-7, far inside the 1e-3 parity bar.
  return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}

// Order this wave's LDS traffic: lanes of one wave exchange data through LDS without s_barrier
// (DS operations of a wave execute in issue order); this only stops the compiler reordering them.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP lane permutes (VALU-rate, no LDS crossbar): quad swaps for distance 1 and 2, row_half_mirror
// and row_mirror pair the two halves of an 8- / 16-lane group -- enough for an all-reduce inside a
// 16-lane row.  Crossing rows (distance 16, 32) goes through ds_bpermute (__shfl_xor).
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

// All-reduce over aligned groups of W lanes (W a power of two); every lane of the group gets the result.
// All lanes of the group must be active.
// Workgroup barrier that only drains LDS traffic (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() also
// waits vmcnt(0), which would expose the latency of global loads deliberately left in flight across the
// barrier (register prefetch of the next round's operands).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int W>
__device__ __forceinline__ float lanes_sum(float v) {
  if constexpr (W >= 2) v += dpp_perm<kDppXor1>(v);
  if constexpr (W >= 4) v += dpp_perm<kDppXor2>(v);
  if constexpr (W >= 8) v += dpp_perm<kDppHalfMirror>(v);
  if constexpr (W >= 16) v += dpp_perm<kDppMirror>(v);
  if constexpr (W >= 32) v += __shfl_xor(v, 16, kWave);
  if constexpr (W >= 64) v += __shfl_xor(v, 32, kWave);
  return v;
}
template <int W>
__device__ __forceinline__ float lanes_max(float v) {
  if constexpr (W >= 2) v = fmaxf(v, dpp_perm<kDppXor1>(v));
  if constexpr (W >= 4) v = fmaxf(v, dpp_perm<kDppXor2>(v));
  if constexpr (W >= 8) v = fmaxf(v, dpp_perm<kDppHalfMirror>(v));
  if constexpr (W >= 16) v = fmaxf(v, dpp_perm<kDppMirror>(v));
  if constexpr (W >= 32) v = fmaxf(v, __shfl_xor(v, 16, kWave));
  if constexpr (W >= 64) v = fmaxf(v, __shfl_xor(v, 32, kWave));
  return v;
}

// Whole-wave reductions that never touch the LDS crossbar: in-row all-reduce, then row_bcast15 /
// row_bcast31 carry the row totals into lane 63, which is read back as a wave-uniform scalar.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_rows(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                               __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xF, false));
}
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143;
__device__ __forceinline__ float wave_sum(float v) {
  v = lanes_sum<16>(v);
  v += dpp_rows<kDppBcast15, 0xA>(0.f, v);
  v += dpp_rows<kDppBcast31, 0xC>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = lanes_max<16>(v);
  v = fmaxf(v, dpp_rows<kDppBcast15, 0xA>(-INFINITY, v));
  v = fmaxf(v, dpp_rows<kDppBcast31, 0xC>(-INFINITY, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

typedef float f2v __attribute__((ext_vector_type(2)));

template <int G_, int VEC_, int NCH_>
struct FeatCfg {
  static constexpr int G = G_;      // lanes per feature row
  static constexpr int VEC = VEC_;  // floats per lane per chunk (4 -> dwordx4 loads, 1 -> dword)
  static constexpr int NCH = NCH_;  // chunks per lane (covers f <= G*VEC*NCH)
  static constexpr int EPW = kWave / G_;  // rows (edges) a wave holds at once
  static_assert(G_ >= 1 && G_ <= 64 && (G_ & (G_ - 1)) == 0, "G must be a power of two <= 64");
  static_assert(VEC_ == 1 || VEC_ == 4, "VEC is 1 or 4");
};

template <class C>
struct Frag {
  float v[C::NCH][C::VEC];
};

template <class C>
__device__ __forceinline__ void frag_zero(Frag<C> &a) {
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch)
#pragma unroll
    for (int k = 0; k < C::VEC; ++k) a.v[ch][k] = 0.f;
}

// Load this lane's slice of the feature row that starts at `base` (already offset to node+head).
template <class C>
__device__ __forceinline__ void frag_load(Frag<C> &a, const float *__restrict__ base, int f, int gl) {
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch) {
    const int c = (ch * C::G + gl) * C::VEC;
    if constexpr (C::VEC == 4) {
      if (c < f) {
        const float4 t = *reinterpret_cast<const float4 *>(base + c);
        a.v[ch][0] = t.x; a.v[ch][1] = t.y; a.v[ch][2] = t.z; a.v[ch][3] = t.w;
      } else {
        a.v[ch][0] = a.v[ch][1] = a.v[ch][2] = a.v[ch][3] = 0.f;
      }
    } else {
      a.v[ch][0] = (c < f) ? base[c] : 0.f;
    }
  }
}

template <class C>
__device__ __forceinline__ void frag_store_scaled(const Frag<C> &a, float s, float *__restrict__ base, int f, int gl) {
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch) {
    const int c = (ch * C::G + gl) * C::VEC;
    if constexpr (C::VEC == 4) {
      if (c < f)
        *reinterpret_cast<float4 *>(base + c) =
            make_float4(a.v[ch][0] * s, a.v[ch][1] * s, a.v[ch][2] * s, a.v[ch][3] * s);
    } else {
      if (c < f) base[c] = a.v[ch][0] * s;
    }
  }
}

template <class C>
__device__ __forceinline__ float frag_dot(const Frag<C> &a, const Frag<C> &b) {
  float d = 0.f;
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch)
#pragma unroll
    for (int k = 0; k < C::VEC; ++k) d = fmaf(a.v[ch][k], b.v[ch][k], d);
  return d;
}

template <class C>
__device__ __forceinline__ void frag_fma(Frag<C> &acc, float w, const Frag<C> &x) {
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch)
#pragma unroll
    for (int k = 0; k < C::VEC; ++k) acc.v[ch][k] = fmaf(w, x.v[ch][k], acc.v[ch][k]);
}

// ---- exact-width forms (f == G*VEC*NCH known at compile time): no lane masks, packed FMAs -----------
template <class C>
__device__ __forceinline__ void frag_load_full(Frag<C> &a, const float *base, int gl) {
  static_assert(C::VEC == 4, "exact-width form is float4 only");
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch) {
    const float4 t = *reinterpret_cast<const float4 *>(base + (ch * C::G + gl) * 4);
    a.v[ch][0] = t.x; a.v[ch][1] = t.y; a.v[ch][2] = t.z; a.v[ch][3] = t.w;
  }
}
template <class C>
__device__ __forceinline__ void frag_store_full(const Frag<C> &a, float s, float *__restrict__ base, int gl) {
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch)
    *reinterpret_cast<float4 *>(base + (ch * C::G + gl) * 4) =
        make_float4(a.v[ch][0] * s, a.v[ch][1] * s, a.v[ch][2] * s, a.v[ch][3] * s);
}
template <class C>
__device__ __forceinline__ float frag_dot_pk(const Frag<C> &a, const Frag<C> &b) {  // v_pk_fma_f32
  f2v acc = {0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch) {
    acc = __builtin_elementwise_fma(f2v{a.v[ch][0], a.v[ch][1]}, f2v{b.v[ch][0], b.v[ch][1]}, acc);
    acc = __builtin_elementwise_fma(f2v{a.v[ch][2], a.v[ch][3]}, f2v{b.v[ch][2], b.v[ch][3]}, acc);
  }
  return acc.x + acc.y;
}
template <class C>
__device__ __forceinline__ void frag_fma_pk(Frag<C> &acc, float w, const Frag<C> &x) {
  const f2v ww = {w, w};
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch) {
    f2v lo = __builtin_elementwise_fma(f2v{x.v[ch][0], x.v[ch][1]}, ww, f2v{acc.v[ch][0], acc.v[ch][1]});
    f2v hi = __builtin_elementwise_fma(f2v{x.v[ch][2], x.v[ch][3]}, ww, f2v{acc.v[ch][2], acc.v[ch][3]});
    acc.v[ch][0] = lo.x; acc.v[ch][1] = lo.y; acc.v[ch][2] = hi.x; acc.v[ch][3] = hi.y;
  }
}

template <class C>
__device__ __forceinline__ void frag_scale(Frag<C> &acc, float s) {
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch)
#pragma unroll
    for (int k = 0; k < C::VEC; ++k) acc.v[ch][k] *= s;
}

// v_permlane{32,16}_swap_b32 through inline asm (hipcc 7.2 mis-selects the second result of the builtin).
// swap32: lanes 32-63 of a <-> lanes 0-31 of b.  swap16: odd 16-lane rows of a <-> even rows of b.
__device__ __forceinline__ void permlane32_swap(float &a, float &b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void permlane16_swap(float &a, float &b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}

// Sum the four per-group partial rows of a wave (G == 16) with 3 swaps + 3 adds per float4 chunk instead of
// 8 ds_bpermute + 8 adds.  On return t[ch] of lane (row r, gl) holds the total of component kGroupSumComp[r]
// of chunk ch at lane position gl, i.e. float (ch*16 + gl)*4 + kGroupSumComp[r] of the feature row.
template <class C>
__device__ __forceinline__ void frag_reduce_groups_swap(Frag<C> &acc, float (&t)[C::NCH]) {
  static_assert(C::G == 16 && C::VEC == 4, "swap reduction is for the 16-lane row layout");
#pragma unroll
  for (int ch = 0; ch < C::NCH; ++ch) {
    float r0 = acc.v[ch][0], r1 = acc.v[ch][1], r2 = acc.v[ch][2], r3 = acc.v[ch][3];
    permlane32_swap(r0, r1);   // r0 = [r0.lo32, r1.lo32], r1 = [r0.hi32, r1.hi32]
    permlane32_swap(r2, r3);
    float s0 = r0 + r1;        // rows 0,1: comp 0 (rows 0+2, 1+3); rows 2,3: comp 1
    float s1 = r2 + r3;        // rows 0,1: comp 2;                 rows 2,3: comp 3
    permlane16_swap(s0, s1);   // s0 = [s0.r0, s1.r0, s0.r2, s1.r2], s1 = [s0.r1, s1.r1, s0.r3, s1.r3]
    t[ch] = s0 + s1;           // row 0: comp 0, row 1: comp 2, row 2: comp 1, row 3: comp 3
  }
}
__device__ __forceinline__ int group_sum_comp(int row) { return ((row & 1) << 1) | (row >> 1); }

// Sum the EPW per-group partial rows of a wave; afterwards every group holds the total.
template <class C>
__device__ __forceinline__ void frag_reduce_groups(Frag<C> &acc) {
#pragma unroll
  for (int o = C::G; o < kWave; o <<= 1)
#pragma unroll
    for (int ch = 0; ch < C::NCH; ++ch)
#pragma unroll
      for (int k = 0; k < C::VEC; ++k) acc.v[ch][k] += __shfl_xor(acc.v[ch][k], o, kWave);
}

// acc += sum_{e<n} wf(w[e]) * X[cols[e]]  where w/cols are wave-visible arrays (LDS or global) and
// the wave's EPW groups stride over the n edges.  4 gathers are kept in flight per group.
template <class C, class WF>
__device__ __forceinline__ void spmm_accum(Frag<C> &acc, const float *w, const int *cols, int n,
                                           const float *__restrict__ X, size_t hf, int f, int gid, int gl, WF wf) {
  int e = gid;
  for (; e + 3 * C::EPW < n; e += 4 * C::EPW) {
    const int c0 = cols[e], c1 = cols[e + C::EPW], c2 = cols[e + 2 * C::EPW], c3 = cols[e + 3 * C::EPW];
    const float w0 = wf(w[e]), w1 = wf(w[e + C::EPW]), w2 = wf(w[e + 2 * C::EPW]), w3 = wf(w[e + 3 * C::EPW]);
    Frag<C> x0, x1, x2, x3;
    frag_load<C>(x0, X + (size_t)c0 * hf, f, gl);
    frag_load<C>(x1, X + (size_t)c1 * hf, f, gl);
    frag_load<C>(x2, X + (size_t)c2 * hf, f, gl);
    frag_load<C>(x3, X + (size_t)c3 * hf, f, gl);
    frag_fma<C>(acc, w0, x0);
    frag_fma<C>(acc, w1, x1);
    frag_fma<C>(acc, w2, x2);
    frag_fma<C>(acc, w3, x3);
  }
  for (; e < n; e += C::EPW) {
    Frag<C> x0;
    frag_load<C>(x0, X + (size_t)cols[e] * hf, f, gl);
    frag_fma<C>(acc, wf(w[e]), x0);
  }
}
template <class C>
__device__ __forceinline__ void spmm_accum(Frag<C> &acc, const float *w, const int *cols, int n,
                                           const float *__restrict__ X, size_t hf, int f, int gid, int gl) {
  spmm_accum<C>(acc, w, cols, n, X, hf, f, gid, gl, [](float x) { return x; });
}

// One online-softmax step for a tile of nt (<= 64) logits, one per lane (lanes >= nt hold -inf).
// Writes the un-normalised probabilities exp(s - m_new) to sw[lane], rescales acc, updates m, l.
template <class C>
__device__ __forceinline__ void online_step(float s, int lane, float *sw, Frag<C> &acc, float &m_run, float &l_run) {
  const float tmax = lanes_max<kWave>(s);
  const float m_new = fmaxf(m_run, tmax);
  // m_run == -inf on the first tile: nothing accumulated yet, the rescale factor is irrelevant (0).
  const float scale = (m_run == -INFINITY) ? 0.f : fast_exp(m_run - m_new);
  const float p = (s == -INFINITY) ? 0.f : fast_exp(s - m_new);
  sw[lane] = p;
  l_run = l_run * scale + lanes_sum<kWave>(p);
  frag_scale<C>(acc, scale);
  m_run = m_new;
}

__device__ __forceinline__ float leaky_relu(float x, float slope) { return x > 0.f ? x : x * slope; }

}  // namespace dfgnn
