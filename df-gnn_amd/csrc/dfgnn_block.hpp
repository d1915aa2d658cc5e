// dfgnn_block.hpp -- device helpers shared by the LDS-resident per-range ("block") kernels
// (gt_block.hip forward, gt_block_bwd.hip backward, gat_block.hip).  See gt_block.hip for the design.
#pragma once
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {


// Diagnostic build only (-DDFGNN_STAMPS, never shipped): per-workgroup phase boundaries in shader cycles.
#ifdef DFGNN_STAMPS
__device__ unsigned long long *dfgnn_stamps = nullptr;
#define DFGNN_STAMP(k)                                                                            \
  if (threadIdx.x == 0 && dfgnn_stamps)                                                           \
    dfgnn_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + (k)] = __builtin_amdgcn_s_memtime();
__device__ unsigned long long *dfgnn_round_stamps = nullptr;   // [wg][8]: per-phase cycle sums of the MFMA rounds
#define DFGNN_RT(var) unsigned long long var = __builtin_amdgcn_s_memtime();
#define DFGNN_RACC(k, a, b) if (threadIdx.x == 0) racc[k] += (b) - (a);
#else
#define DFGNN_STAMP(k)
#define DFGNN_RT(var)
#define DFGNN_RACC(k, a, b)
#endif

// Per-edge fp32 values of a range (logits / exp values / dS): in LDS when the range's plan entry allows,
// otherwise in caller-provided global scratch (attn_edge in the training forward, grad_edge in the
// backward).  Every access is block-uniformly one or the other, so no flat-address instructions are needed.
struct EdgeArr {
  float *lds;      // null when the array lives in global memory
  float *glob;
  __device__ __forceinline__ float load(int e) const { return lds ? lds[e] : glob[e]; }
  __device__ __forceinline__ void store(int e, float v) const {
    if (lds) lds[e] = v; else glob[e] = v;
  }
};

struct BlockLds {
  float *res;            // [n * f]  resident feature rows (K, then V)
  float *lw;             // [ne]     raw logits, then exp(s - max)   (zero-sized when the edge array is global)
  float *rinv;           // [n]      1 / row sum
  int *rp;               // [n + 1]  row_ptr of the range, relative to its first edge
  unsigned char *cols;   // [ne]     block-local column ids, 1 byte each if n <= 256 else 2 bytes
  int2 *sc;              // this wave's 64 x (col, weight) staging
};

// Layout must stay in step with plan.hip:bytes_of() (the plan guarantees it fits 160 KB).
__device__ __forceinline__ BlockLds carve_block_lds(float *lds, int n, int ne, int f, int wave,
                                                    int res_floats = -1) {
  BlockLds b;
  b.res = lds;
  b.lw = b.res + (res_floats >= 0 ? (size_t)res_floats : (size_t)n * f);
  b.rinv = b.lw + ((ne + 3) & ~3);
  b.rp = reinterpret_cast<int *>(b.rinv + ((n + 3) & ~3));
  b.sc = reinterpret_cast<int2 *>(b.rp + ((n + 1 + 3) & ~3)) + wave * kWave;
  b.cols = reinterpret_cast<unsigned char *>(reinterpret_cast<int2 *>(b.rp + ((n + 1 + 3) & ~3)) + kBlockWaves * kWave);
  return b;
}

// Stage the range's row_ptr (rebased) and column ids (rebased, narrowed) into LDS.
__device__ __forceinline__ void load_block_index(const BlockLds &L, const Csr &g, int n0, int n, int e0, int ne) {
  for (int i = threadIdx.x; i <= n; i += kBlockThreads) L.rp[i] = g.row_ptr[n0 + i] - e0;
  const int *ci = g.col_ind + e0;
  if (n <= 256) {
    for (int b = threadIdx.x * 4; b < ne; b += kBlockThreads * 4) {
      unsigned v = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (b + k < ne) v |= (unsigned)(ci[b + k] - n0) << (8 * k);
      *reinterpret_cast<unsigned *>(L.cols + b) = v;
    }
  } else {
    unsigned short *c16 = reinterpret_cast<unsigned short *>(L.cols);
    for (int b = threadIdx.x * 2; b < ne; b += kBlockThreads * 2) {
      unsigned v = (unsigned)(ci[b] - n0);
      if (b + 1 < ne) v |= (unsigned)(ci[b + 1] - n0) << 16;
      *reinterpret_cast<unsigned *>(c16 + b) = v;
    }
  }
}

__device__ __forceinline__ int block_col(const BlockLds &L, bool narrow, int e) {
  return narrow ? (int)L.cols[e] : (int)reinterpret_cast<const unsigned short *>(L.cols)[e];
}

// Copy n feature rows (f floats each, row stride hf in global memory) into LDS, float4 per lane.
__device__ __forceinline__ void load_resident(float *res, const float *__restrict__ src, int n, int f, size_t hf) {
  const int f4 = f >> 2;
  const int total = n * f4;
  float4 *dst4 = reinterpret_cast<float4 *>(res);
  if (hf == (size_t)f) {
    const float4 *src4 = reinterpret_cast<const float4 *>(src);
    int idx = threadIdx.x;
    for (; idx + 3 * kBlockThreads < total; idx += 4 * kBlockThreads) {
      const float4 a = src4[idx], b = src4[idx + kBlockThreads], c = src4[idx + 2 * kBlockThreads],
                   d = src4[idx + 3 * kBlockThreads];
      dst4[idx] = a;
      dst4[idx + kBlockThreads] = b;
      dst4[idx + 2 * kBlockThreads] = c;
      dst4[idx + 3 * kBlockThreads] = d;
    }
    for (; idx < total; idx += kBlockThreads) dst4[idx] = src4[idx];
  } else {
    for (int idx = threadIdx.x; idx < total; idx += kBlockThreads) {
      const int row = idx / f4, c = idx - row * f4;
      dst4[idx] = *reinterpret_cast<const float4 *>(src + (size_t)row * hf + 4 * c);
    }
  }
}

// Which edge of a 64-edge chunk lane (gid, gl) ends up holding after block_chunk_logits.
template <class C>
__device__ __forceinline__ int chunk_edge_of_lane(int gid, int gl) {
  if constexpr (C::G == 16) return (4 * (gl & 3) + (gl >> 2)) * C::EPW + gid;  // reduce-scatter layout
  return gl * C::EPW + gid;
}

// Logits of one <= 64-edge chunk of a row: group gid handles the chunk's edges gid, gid+EPW, ... (their
// block-local columns were staged de-interleaved in sci); on return lane (gid, gl) holds the logit of
// edge chunk_edge_of_lane(gid, gl) of the chunk.  Four iterations (4*EPW edges, 4*NCH LDS row reads in
// flight) per trip; slots past the chunk hold column 0, so surplus iterations read a valid row.
// G == 16: the four partial dot products of a trip are reduce-scattered over the 16 lanes (row_mirror,
// row_half_mirror, two quad swaps: 11 VALU ops, quad j ends with the total of iteration it + j) instead of
// four all-reduces (16 ops + 4 selects).
template <class C>
__device__ __forceinline__ float block_chunk_logits(const float *res, const int *sci, const Frag<C> &q, int nt,
                                                    int gid, int gl) {
  constexpr int G = C::G, EPW = C::EPW, F = C::G * C::VEC * C::NCH;
  const int iters = (nt + EPW - 1) / EPW;
  float mine = 0.f;
  const bool upper = (gl & 8) != 0, odd4 = (gl & 4) != 0;
  for (int it = 0; it < iters; it += 4) {
    const int4 c4 = *reinterpret_cast<const int4 *>(sci + gid * G + it);
    Frag<C> k0, k1, k2, k3;
    frag_load_full<C>(k0, res + c4.x * F, gl);
    frag_load_full<C>(k1, res + c4.y * F, gl);
    frag_load_full<C>(k2, res + c4.z * F, gl);
    frag_load_full<C>(k3, res + c4.w * F, gl);
    float d0 = frag_dot_pk<C>(q, k0), d1 = frag_dot_pk<C>(q, k1), d2 = frag_dot_pk<C>(q, k2),
          d3 = frag_dot_pk<C>(q, k3);
    if constexpr (G == 16) {
      const float x0 = (upper ? d2 : d0) + dpp_perm<kDppMirror>(upper ? d0 : d2);
      const float x1 = (upper ? d3 : d1) + dpp_perm<kDppMirror>(upper ? d1 : d3);
      float y = (odd4 ? x1 : x0) + dpp_perm<kDppHalfMirror>(odd4 ? x0 : x1);
      y += dpp_perm<kDppXor2>(y);
      y += dpp_perm<kDppXor1>(y);
      asm volatile("" : "+v"(y));  // keep the select below a v_cndmask
      mine = ((gl & 3) == (it >> 2)) ? y : mine;
    } else {
      d0 = lanes_sum<G>(d0);
      d1 = lanes_sum<G>(d1);
      d2 = lanes_sum<G>(d2);
      d3 = lanes_sum<G>(d3);
      asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));  // keep the selects below as v_cndmask
      const int rel = gl - it;
      mine = rel == 0 ? d0 : mine;
      mine = rel == 1 ? d1 : mine;
      mine = rel == 2 ? d2 : mine;
      mine = rel == 3 ? d3 : mine;
    }
  }
  return mine;
}

// out_row = scale * (sum of the wave's per-group partial rows): the epilogue of every SpMM pass.
template <class C>
__device__ __forceinline__ void block_store_row(Frag<C> &acc, float scale, float *__restrict__ out_row, int gid,
                                                int gl) {
  if constexpr (C::G == 16) {
    float t[C::NCH];
    frag_reduce_groups_swap<C>(acc, t);
    const int comp = group_sum_comp(gid);
#pragma unroll
    for (int ch = 0; ch < C::NCH; ++ch) out_row[(ch * 16 + gl) * 4 + comp] = t[ch] * scale;
  } else {
    frag_reduce_groups<C>(acc);
    if (gid == 0) frag_store_full<C>(acc, scale, out_row, gl);
  }
}

// acc += sum_{k<deg} w_k * res[row_k]: the entries of one CSR row (or CSC column) are produced by
// `entry(k, row, w)` on lane k % 64 of each 64-entry chunk, staged de-interleaved in the wave's scratch and
// consumed by the EPW lane groups, four iterations (4*EPW entries, 4*NCH LDS reads) per trip.
template <class C, class EntryFn>
__device__ __forceinline__ void block_spmm(Frag<C> &acc, const float *res, int2 *sc, int deg, int lane, int gid,
                                           int gl, EntryFn entry) {
  constexpr int G = C::G, EPW = C::EPW, F = C::G * C::VEC * C::NCH;
  const int stage = (lane % EPW) * G + lane / EPW;
  for (int c0 = 0; c0 < deg; c0 += kWave) {
    const int nt = min(kWave, deg - c0);
    float w = 0.f;
    int row = 0;
    if (lane < nt) entry(c0 + lane, row, w);
    sc[stage] = make_int2(row * F, __float_as_int(w));
    wave_sync();
    const int iters = (nt + EPW - 1) / EPW;
    constexpr int TRIP = 4;
    for (int it = 0; it < iters; it += TRIP) {
      int ro[TRIP];
      float wv[TRIP];
#pragma unroll
      for (int u = 0; u < TRIP; u += 2) {
        const int4 a = *reinterpret_cast<const int4 *>(sc + gid * G + it + u);  // (row offset, w) x 2
        ro[u] = a.x; wv[u] = __int_as_float(a.y); ro[u + 1] = a.z; wv[u + 1] = __int_as_float(a.w);
      }
      Frag<C> v[TRIP];
#pragma unroll
      for (int u = 0; u < TRIP; ++u) frag_load_full<C>(v[u], res + ro[u], gl);
#pragma unroll
      for (int u = 0; u < TRIP; ++u) frag_fma_pk<C>(acc, wv[u], v[u]);
    }
    wave_sync();
  }
}

typedef __attribute__((ext_vector_type(4))) float f32x4;

size_t block_lds_bytes(const Plan &p, int f);

// Raise a kernel's dynamic-LDS ceiling so launches above 64 KB are accepted.
// (once per kernel and device: the launchers call this on every launch, the attribute call itself is remembered --
// dfgnn_launch.hpp: set_max_lds_cached_ptr)
template <class K>
static int set_max_lds_cached(K kernel) {
  return set_max_lds_cached_ptr(reinterpret_cast<const void *>(kernel));
}

}  // namespace dfgnn
