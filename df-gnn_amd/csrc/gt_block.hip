// gt_block.hip -- the MI355X-native 'hyper' kernels for batched (block-diagonal) graphs.
//
// One workgroup of 16 waves owns one closed node range of the block plan (plan.hip; for a DGL batch:
// one member graph, or a few merged small ones) and one head.  The range's K rows are copied into
// LDS once, every <Q_i, K_j> of the range is then served from LDS; the raw logits stay in LDS; the
// K rows are replaced by the V rows and the softmax-weighted sum is again served from LDS.  HBM sees
// each of Q, K, V, the index arrays and the output exactly once; the 512-byte-per-edge gathers that
// bound the reference kernels (fused_gtconv_hyper.cu:333-337, 399-409 -- left to the L2 there) never
// leave the CU.
//
// Lane layout (FeatCfg): a feature row is held by G <= 16 lanes (one DPP row), so a wave touches
// EPW = 64/G edges per LDS instruction and the dot-product reduction is 4 VALU-rate DPP adds.
// Within a 64-edge chunk, edge k is handled by lane group k % EPW in iteration k / EPW; the chunk's
// (column, weight) pairs are staged de-interleaved in a 512-byte per-wave scratch so that a group reads
// its next column with one broadcast LDS read.
// Rows are dealt to the 16 waves round-robin (an LDS ticket counter would balance better, but hipcc 7.2
// mis-structures the `for(;;){ticket; if (r>=n) break; ...}` loop around an aggregated LDS atomic: the
// kernel never terminated on hardware).
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

// Diagnostic build only (-DDFGNN_STAMPS, never shipped): per-workgroup phase boundaries in shader cycles.
#ifdef DFGNN_STAMPS
__device__ unsigned long long *dfgnn_stamps = nullptr;
#define DFGNN_STAMP(k)                                                                            \
  if (threadIdx.x == 0 && dfgnn_stamps)                                                           \
    dfgnn_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define DFGNN_STAMP(k)
#endif

struct BlockLds {
  float *res;            // [n * f]  resident feature rows (K, then V)
  float *lw;             // [ne]     raw logits, then exp(s - max)
  float *rinv;           // [n]      1 / row sum
  int *rp;               // [n + 1]  row_ptr of the range, relative to its first edge
  unsigned char *cols;   // [ne]     block-local column ids, 1 byte each if n <= 256 else 2 bytes
  int2 *sc;              // this wave's 64 x (col, weight) staging
};

// Layout must stay in step with plan.hip:bytes_of() (the plan guarantees it fits 160 KB).
__device__ __forceinline__ BlockLds carve_block_lds(float *lds, int n, int ne, int f, int wave) {
  BlockLds b;
  b.res = lds;
  b.lw = b.res + (size_t)n * f;
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

// Logits of one <= 64-edge chunk of a row: group gid handles the chunk's edges gid, gid+EPW, ... (their
// block-local columns were staged de-interleaved in sci); on return lane (gid, gl) holds the logit of
// edge gl*EPW + gid of the chunk.  Four iterations (4*EPW edges) per trip keep 4*NCH LDS reads in flight.
template <class C>
__device__ __forceinline__ float block_chunk_logits(const float *res, const int *sci, const Frag<C> &q, int nt,
                                                    int gid, int gl) {
  constexpr int G = C::G, EPW = C::EPW, F = C::G * C::VEC * C::NCH;
  const int iters = (nt + EPW - 1) / EPW;
  float mine = 0.f;
  for (int it = 0; it < iters; it += 4) {
    const int4 c4 = *reinterpret_cast<const int4 *>(sci + gid * G + it);  // slots past the chunk hold col 0
    Frag<C> k0, k1, k2, k3;
    frag_load_full<C>(k0, res + c4.x * F, gl);
    frag_load_full<C>(k1, res + c4.y * F, gl);
    frag_load_full<C>(k2, res + c4.z * F, gl);
    frag_load_full<C>(k3, res + c4.w * F, gl);
    float d0 = lanes_sum<G>(frag_dot_pk<C>(q, k0));
    float d1 = lanes_sum<G>(frag_dot_pk<C>(q, k1));
    float d2 = lanes_sum<G>(frag_dot_pk<C>(q, k2));
    float d3 = lanes_sum<G>(frag_dot_pk<C>(q, k3));
    asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));  // keep the selects below as v_cndmask
    const int rel = gl - it;
    mine = rel == 0 ? d0 : mine;
    mine = rel == 1 ? d1 : mine;
    mine = rel == 2 ? d2 : mine;
    mine = rel == 3 ? d3 : mine;
  }
  return mine;
}

template <class C, bool WRITE_ATTN>
__global__ __launch_bounds__(kBlockThreads) void gt_block_fwd_kernel(Csr g, const int *__restrict__ fit,
                                                                     const float *__restrict__ Q,
                                                                     const float *__restrict__ K,
                                                                     const float *__restrict__ V,
                                                                     float *__restrict__ attn_edge,
                                                                     float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int G = C::G, EPW = C::EPW, F = C::G * C::VEC * C::NCH;  // F == g.f (checked by the launcher)
  const int n0 = fit[2 * blockIdx.x], n1 = fit[2 * blockIdx.x + 1];
  const int n = n1 - n0;
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * F;
  const int e0 = g.row_ptr[n0];
  const int ne = g.row_ptr[n1] - e0;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int gid = lane / G, gl = lane % G;
  const BlockLds L = carve_block_lds(lds, n, ne, F, wave);
  int *sci = reinterpret_cast<int *>(L.sc);  // pass A stages columns only (first 256 B of the scratch)
  const float *Qh = Q + (size_t)head * F;
  float *outh = out + (size_t)head * F;
  float *attn_h = WRITE_ATTN ? attn_edge + (size_t)head * g.nnz : nullptr;
  const bool narrow = n <= 256;
  const int kk = gl * EPW + gid;            // the edge of a chunk whose logit this lane keeps
  const int stage = (lane % EPW) * G + lane / EPW;  // where lane's own edge goes in the de-interleaved scratch

  DFGNN_STAMP(0)
  load_block_index(L, g, n0, n, e0, ne);
  load_resident(L.res, K + (size_t)n0 * hf + (size_t)head * F, n, F, hf);
  __syncthreads();
  DFGNN_STAMP(1)

  // ---- pass A: logits of every edge of the range, K served from LDS; then the row softmax ----------
  Frag<C> q_next;
  if (wave < n) frag_load_full<C>(q_next, Qh + (size_t)(n0 + wave) * hf, gl);
  for (int r = wave; r < n; r += kBlockWaves) {
    // block-relative edge range of row n0 + r (wave-uniform: keep the loop control in SGPRs)
    const int lb = __builtin_amdgcn_readfirstlane(L.rp[r]);
    const int deg = __builtin_amdgcn_readfirstlane(L.rp[r + 1]) - lb;
    const Frag<C> q = q_next;
    if (r + kBlockWaves < n) frag_load_full<C>(q_next, Qh + (size_t)(n0 + r + kBlockWaves) * hf, gl);
    float *lrow = L.lw + lb;
    if (deg <= kWave) {
      // the whole row is one chunk: softmax entirely in registers
      const float myval = (g.val && kk < deg) ? g.val[e0 + lb + kk] : 1.f;
      sci[stage] = (lane < deg) ? block_col(L, narrow, lb + lane) : 0;
      wave_sync();
      const float mine = block_chunk_logits<C>(L.res, sci, q, deg, gid, gl);
      wave_sync();
      const float s = (kk < deg) ? mine * myval : -INFINITY;
      const float mx = wave_max(s);
      const float p = (s == -INFINITY) ? 0.f : fast_exp(s - mx);
      const float sum = wave_sum(p);
      if (kk < deg) lrow[kk] = p;
      if (lane == 0) L.rinv[r] = (sum != 0.f) ? 1.f / sum : 0.f;
    } else {
      for (int c0 = 0; c0 < deg; c0 += kWave) {
        const int nt = min(kWave, deg - c0);
        const float myval = (g.val && kk < nt) ? g.val[e0 + lb + c0 + kk] : 1.f;
        sci[stage] = (lane < nt) ? block_col(L, narrow, lb + c0 + lane) : 0;
        wave_sync();
        const float mine = block_chunk_logits<C>(L.res, sci, q, nt, gid, gl);
        wave_sync();
        if (kk < nt) lrow[c0 + kk] = mine * myval;
      }
      wave_sync();
      float mx = -INFINITY;
      for (int e = lane; e < deg; e += kWave) mx = fmaxf(mx, lrow[e]);
      mx = wave_max(mx);
      float sum = 0.f;
      for (int e = lane; e < deg; e += kWave) {
        const float s = lrow[e];
        const float p = (s == -INFINITY) ? 0.f : fast_exp(s - mx);
        lrow[e] = p;
        sum += p;
      }
      sum = wave_sum(sum);
      if (lane == 0) L.rinv[r] = (sum != 0.f) ? 1.f / sum : 0.f;
    }
  }
  DFGNN_STAMP(2)
  __syncthreads();
  DFGNN_STAMP(3)

  // ---- swap the resident rows: K -> V ----------------------------------------------------------------
  load_resident(L.res, V + (size_t)n0 * hf + (size_t)head * F, n, F, hf);
  __syncthreads();
  DFGNN_STAMP(4)

  // ---- pass B: out_i = (1/sum_i) * sum_e exp_e V_j, V served from LDS ---------------------------------
  for (int r = wave; r < n; r += kBlockWaves) {
    const int i = n0 + r;
    const int lb = __builtin_amdgcn_readfirstlane(L.rp[r]);
    const int deg = __builtin_amdgcn_readfirstlane(L.rp[r + 1]) - lb;
    const float inv = L.rinv[r];
    const float *lrow = L.lw + lb;
    Frag<C> acc;
    frag_zero<C>(acc);
    for (int c0 = 0; c0 < deg; c0 += kWave) {
      const int nt = min(kWave, deg - c0);
      float w = 0.f;
      int col = 0;
      if (lane < nt) {
        w = lrow[c0 + lane];
        col = block_col(L, narrow, lb + c0 + lane);
        if constexpr (WRITE_ATTN) attn_h[e0 + lb + c0 + lane] = w * inv;
      }
      L.sc[stage] = make_int2(col * F, __float_as_int(w));
      wave_sync();
      const int iters = (nt + EPW - 1) / EPW;
      for (int it = 0; it < iters; it += 4) {
        const int4 a = *reinterpret_cast<const int4 *>(L.sc + gid * G + it);       // (row offset, w) x 2
        const int4 b = *reinterpret_cast<const int4 *>(L.sc + gid * G + it + 2);   // (row offset, w) x 2
        Frag<C> v0, v1, v2, v3;
        frag_load_full<C>(v0, L.res + a.x, gl);
        frag_load_full<C>(v1, L.res + a.z, gl);
        frag_load_full<C>(v2, L.res + b.x, gl);
        frag_load_full<C>(v3, L.res + b.z, gl);
        frag_fma_pk<C>(acc, __int_as_float(a.y), v0);
        frag_fma_pk<C>(acc, __int_as_float(a.w), v1);
        frag_fma_pk<C>(acc, __int_as_float(b.y), v2);
        frag_fma_pk<C>(acc, __int_as_float(b.w), v3);
      }
      wave_sync();
    }
    frag_reduce_groups<C>(acc);
    if (gid == 0) frag_store_full<C>(acc, inv, outh + (size_t)i * hf, gl);
  }
  DFGNN_STAMP(5)
  if (threadIdx.x == 0) { DFGNN_STAMP(6) }
#ifdef DFGNN_STAMPS
  if (threadIdx.x == 0 && dfgnn_stamps) dfgnn_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + 7] = ((unsigned long long)n << 32) | (unsigned)ne;
#endif
}

static size_t block_lds_bytes(const Plan &p, int f) {
  const size_t n = p.max_fit_nodes, e = p.max_fit_edges;
  size_t b = n * f * 4 + (e + 4) * 4 + (n + 4) * 4 + (n + 8) * 4 + kBlockScratchBytes + (e + 4) * 2 + 64;
  return b > (size_t)kLdsBytes ? (size_t)kLdsBytes : b;
}

// Raise the kernel's dynamic-LDS ceiling (once per size) so launches above 64 KB are accepted.
template <class K>
static int set_max_lds(K kernel, size_t bytes) {
  static size_t granted = 0;  // one instance per kernel instantiation (K is a distinct function type per use)
  static const void *granted_for = nullptr;
  const void *fn = reinterpret_cast<const void *>(kernel);
  if (granted_for == fn && bytes <= granted) return 0;
  const int rc = (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  if (rc == 0) { granted = kLdsBytes; granted_for = fn; }
  return rc;
}

// The resident kernels are compiled for exact row widths (f == G*VEC*NCH: 16, 32, 64, 128, 256, 512, 1024).
bool block_width_ok(int f) {
  return dispatch_vec4(f, [&](auto cfg) {
    using C = decltype(cfg);
    return (int)(C::G * C::VEC * C::NCH == f);
  }) == 1;
}

int launch_gt_block_fwd(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                        float *attn_edge, float *out, hipStream_t s) {
  if (p.num_fit == 0) return 0;
  const dim3 grid(p.num_fit, g.h);
  const size_t lds = block_lds_bytes(p, g.f);
  return dispatch_vec4(g.f, [&](auto cfg) {
    using C = decltype(cfg);
    if (attn_edge) {
      if (int rc = set_max_lds(gt_block_fwd_kernel<C, true>, lds)) return rc;
      gt_block_fwd_kernel<C, true><<<grid, kBlockThreads, lds, s>>>(g, p.fit(), Q, K, V, attn_edge, out);
    } else {
      if (int rc = set_max_lds(gt_block_fwd_kernel<C, false>, lds)) return rc;
      gt_block_fwd_kernel<C, false><<<grid, kBlockThreads, lds, s>>>(g, p.fit(), Q, K, V, nullptr, out);
    }
    return launch_status();
  });
}

}  // namespace dfgnn

#ifdef DFGNN_STAMPS
extern "C" int dfgnn_debug_set_stamps(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_stamps), &p, sizeof(p));
}
#endif
