// dfgnn_dense.hpp -- matrix-core ("dense") form of the per-range GT kernels.
//
// A closed node range of a batched graph is a small square attention problem: n <= 255 nodes, and for the
// dense graphs the headline workload is made of (PATTERN: ~119 nodes, 44 % of all pairs are edges) most of the
// n x n logits are real.  Computing ALL of them on the matrix cores and masking costs ~0.1 cycle per node pair
// per CU; walking the edges on the VALU costs ~7 cycles per edge.  So whenever a range has at least one edge per
// 32 node pairs the whole convolution is done as masked dense attention:
//
//   S^T = K Q^T          v_mfma_f32_16x16x32_bf16 on split-bf16 operands (x = hi + lo; hi*hi + hi*lo + lo*hi
//                        accumulated in fp32, relative error ~2^-16 -- inside the 1e-3 parity bar by 60x)
//   P   = softmax(mask)  in registers: the S^T accumulator layout puts a row of S on the 4 lanes {i, i+16, i+32,
//                        i+48}; the mask is a byte map [i][j] -> position of edge (i, j) in row i (0xFF: no edge),
//                        built once per range in LDS, which also tells where P_ij goes in attn_edge
//   O^T = V^T P^T        the S^T accumulators ARE the B operand of this product (k-slot (q, t) of a 32-column
//                        block <-> column 4q + t, 16 + 4q + t - 4), V^T fragments come from the row-major bf16 V
//                        image through ds_read_b64_tr_b16; the O^T accumulator holds 4 consecutive features of
//                        one output row per lane -> float4 stores
//
// One wave owns 16 rows of the range (a "strip"); K (then V) is staged 128 rows at a time as bf16 hi/lo images
// with a 16-element row skew (conflict-free for both the b128 row reads and the transposed reads).
// Conditions (anything else takes the edge-walking kernels): unit edge values (val == NULL), no duplicate
// edges (checked while the map is built), n <= 255, f in {32, 64, 128}.
#pragma once
#include "dfgnn_block.hpp"

namespace dfgnn {

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) bf16x4 *lds_bf16x4_ptr;

constexpr int kDenseChunkRows = 128;  // rows of K / V resident at a time
constexpr int kDenseMaxNodes = 255;   // edge positions within a row fit a byte, 0xFF = no edge

template <int F>
struct DenseCfg {
  static constexpr bool ok = (F % 32 == 0) && (F <= 128);
  static constexpr int RS = F + 16;  // bf16 elements per image row
  static constexpr int KT = F / 32;  // MFMA k-steps across the feature dimension
  static constexpr int FT = F / 16;  // 16-feature tiles of the output
};

// image (hi + lo) | byte map | row_ptr of the range | counter        (+ optionally ne floats: attn_edge staging)
__host__ __device__ inline size_t dense_lds_bytes(int n, int f) {
  const int npad = (n + 31) & ~31, nstrip = (n + 15) >> 4;
  return (size_t)kDenseChunkRows * (f + 16) * 4 + (size_t)nstrip * 16 * (npad + 4) + (size_t)((n + 4) & ~3) * 4 + 64;
}

// The dense form pays off from about one edge per 32 node pairs (see the header comment).
__host__ __device__ inline bool dense_worthwhile(int n, int ne) { return (long)ne * 32 >= (long)n * n; }

// Stage rows [row0, row0 + rows) of a feature matrix (row 0 = first node of the range, this head; global row
// stride hf floats) as bf16 hi / lo images; rows at or past n are zero.
template <int F>
__device__ __forceinline__ void dense_stage(__bf16 *hi, __bf16 *lo, const float *__restrict__ src, size_t hf, int row0,
                                            int rows, int n) {
  constexpr int C8 = F / 8, RS = DenseCfg<F>::RS;
  for (int idx = threadIdx.x; idx < rows * C8; idx += kBlockThreads) {
    const int row = idx / C8, c8 = idx - row * C8;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (row0 + row < n) {
      const float4 *p = reinterpret_cast<const float4 *>(src + (size_t)(row0 + row) * hf + 8 * c8);
      a = p[0];
      b = p[1];
    }
    bf16x8 h, l;
    split_bf16x8(a, b, h, l);
    *reinterpret_cast<bf16x8 *>(hi + (size_t)row * RS + 8 * c8) = h;
    *reinterpret_cast<bf16x8 *>(lo + (size_t)row * RS + 8 * c8) = l;
  }
}

// Two transposed 4-row reads -> one 8-element operand fragment (rows r .. r+3 and r+16 .. r+19 of a column).
__device__ __forceinline__ bf16x8 dense_tr_pair(const __bf16 *p, int second_offset) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p));
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(p + second_offset));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float xor16_32_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xor16_32_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}

// Byte map [i][j] -> position of edge (i, j) within row i, in LDS, plus the range's row_ptr (rebased to its first
// edge).  Returns false (block-uniformly) when the range has duplicate edges or a row too long for a byte.  `map`
// rows are MS bytes apart; rows / columns past n stay 0xFF.  The caller issues its own long-latency loads first:
// the first kDensePre edges of every thread are fetched before the first barrier.
constexpr int kDensePre = 8;

struct DenseEdgePrefetch {
  int i[kDensePre], j[kDensePre];
};

__device__ __forceinline__ void dense_prefetch_edges(DenseEdgePrefetch &pre, const Csr &g, int e0, int ne) {
#pragma unroll
  for (int k = 0; k < kDensePre; ++k) {
    const int e = threadIdx.x + k * kBlockThreads;
    pre.i[k] = (e < ne) ? g.rows[e0 + e] : 0;
    pre.j[k] = (e < ne) ? g.col_ind[e0 + e] : 0;
  }
}

__device__ __forceinline__ bool dense_build_map(unsigned char *map, int *rp, int *cnt, int map_bytes, int MS,
                                                const DenseEdgePrefetch &pre, const Csr &g, int n0, int n, int e0,
                                                int ne) {
  for (int i = threadIdx.x; i < (map_bytes >> 2); i += kBlockThreads) reinterpret_cast<unsigned *>(map)[i] = 0xFFFFFFFFu;
  int poison = 0;
  for (int i = threadIdx.x; i <= n; i += kBlockThreads) {
    const int a = g.row_ptr[n0 + i];
    rp[i] = a - e0;
    if (i < n && g.row_ptr[n0 + i + 1] - a > 255) poison = 1 << 20;  // a row longer than 255 has duplicates
  }
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kDensePre; ++k) {
    const int e = threadIdx.x + k * kBlockThreads;
    if (e < ne) {
      const int i = pre.i[k] - n0, j = pre.j[k] - n0;
      map[i * MS + j] = (unsigned char)(e - rp[i]);
    }
  }
  for (int e = threadIdx.x + kDensePre * kBlockThreads; e < ne; e += kBlockThreads) {
    const int i = g.rows[e0 + e] - n0, j = g.col_ind[e0 + e] - n0;
    map[i * MS + j] = (unsigned char)(e - rp[i]);
  }
  __syncthreads();
  // every edge must have left its own byte: count them
  int mine = poison;
  for (int i = threadIdx.x; i < (map_bytes >> 2); i += kBlockThreads) {
    const unsigned w = reinterpret_cast<const unsigned *>(map)[i];
#pragma unroll
    for (int b = 0; b < 4; ++b) mine += (((w >> (8 * b)) & 0xFFu) != 0xFFu) ? 1 : 0;
  }
  mine = (int)wave_sum((float)mine) ;
  if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(cnt, mine);
  __syncthreads();
  const bool ok = *cnt == ne;
  if (!ok) __syncthreads();  // the caller re-carves LDS
  return ok;
}

// Registers <-> bf16 images, in two halves so that the global loads of the next image can be in flight while the
// matrix cores work on the current one (the loads are issued before a phase, the split + LDS stores follow it).
template <int F>
struct DenseStageRegs {
  static constexpr int PER = (kDenseChunkRows * (F / 8) + kBlockThreads - 1) / kBlockThreads;  // 16-byte pairs per thread
  float4 a[PER], b[PER];
};

template <int F>
__device__ __forceinline__ void dense_stage_load(DenseStageRegs<F> &r, const float *__restrict__ src, size_t hf, int row0,
                                                 int rows, int n) {
  constexpr int C8 = F / 8;
#pragma unroll
  for (int k = 0; k < DenseStageRegs<F>::PER; ++k) {
    const int idx = threadIdx.x + k * kBlockThreads;
    const int row = idx / C8, c8 = idx - row * C8;
    r.a[k] = r.b[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < rows * C8 && row0 + row < n) {
      const float4 *p = reinterpret_cast<const float4 *>(src + (size_t)(row0 + row) * hf + 8 * c8);
      r.a[k] = p[0];
      r.b[k] = p[1];
    }
  }
}

template <int F>
__device__ __forceinline__ void dense_stage_store(const DenseStageRegs<F> &r, __bf16 *hi, __bf16 *lo, int rows) {
  constexpr int C8 = F / 8, RS = DenseCfg<F>::RS;
#pragma unroll
  for (int k = 0; k < DenseStageRegs<F>::PER; ++k) {
    const int idx = threadIdx.x + k * kBlockThreads;
    const int row = idx / C8, c8 = idx - row * C8;
    if (idx < rows * C8) {
      bf16x8 h, l;
      split_bf16x8(r.a[k], r.b[k], h, l);
      *reinterpret_cast<bf16x8 *>(hi + (size_t)row * RS + 8 * c8) = h;
      *reinterpret_cast<bf16x8 *>(lo + (size_t)row * RS + 8 * c8) = l;
    }
  }
}

// Forward of one range.  NCHUNK = 1: n <= 128, 2: n <= 255.  Returns false, having written nothing, when the
// range turns out not to qualify (duplicate edges).
template <int F, bool WRITE_ATTN, int NCHUNK>
__device__ __forceinline__ bool gt_block_fwd_dense(float *lds, int lds_bytes, const Csr &g, int n0, int n, int e0,
                                                   int ne, int head, const float *__restrict__ Q,
                                                   const float *__restrict__ K, const float *__restrict__ V,
                                                   float *__restrict__ attn_edge, float *__restrict__ out) {
  using D = DenseCfg<F>;
  constexpr int RS = D::RS, KT = D::KT, FT = D::FT, NT = 8 * NCHUNK, CR = kDenseChunkRows;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int mi = lane & 15, mq = lane >> 4;
  const int npad = (n + 31) & ~31, ntile = npad >> 4, nstrip = (n + 15) >> 4;
  const int MS = npad + 4;
  __bf16 *ihi = reinterpret_cast<__bf16 *>(lds), *ilo = ihi + (size_t)CR * RS;
  unsigned char *map = reinterpret_cast<unsigned char *>(ilo + (size_t)CR * RS);
  const int map_bytes = nstrip * 16 * MS;
  int *rp = reinterpret_cast<int *>(map + ((map_bytes + 15) & ~15));
  int *cnt = rp + ((n + 4) & ~3);
  float *pstage = reinterpret_cast<float *>(cnt + 4);  // [ne] normalised attention values in CSR order, if it fits
  const bool stage_attn = WRITE_ATTN && (dense_lds_bytes(n, F) + (size_t)ne * 4 <= (size_t)lds_bytes);
  const size_t hf = (size_t)g.h * F;
  const float *Kb = K + (size_t)n0 * hf + (size_t)head * F;
  const float *Vb = V + (size_t)n0 * hf + (size_t)head * F;
  const bool active = wave < nstrip;  // wave-uniform
  const int i = wave * 16 + mi;       // this lane's row of the range (>= n possible in the last strip)

  DFGNN_STAMP(0)
  // every long-latency load of the prologue goes out before the first barrier
  DenseEdgePrefetch pre;
  dense_prefetch_edges(pre, g, e0, ne);
  DenseStageRegs<F> st;
  dense_stage_load<F>(st, Kb, hf, 0, min(npad, CR), n);
  bf16x8 qh[KT], ql[KT];
  if (active) {
    const float *qrow = Q + (size_t)(n0 + min(i, n - 1)) * hf + (size_t)head * F + 8 * mq;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      float4 a = *reinterpret_cast<const float4 *>(qrow + 32 * t), b = *reinterpret_cast<const float4 *>(qrow + 32 * t + 4);
      if (i >= n) a = b = make_float4(0.f, 0.f, 0.f, 0.f);
      split_bf16x8(a, b, qh[t], ql[t]);
    }
  }
  dense_stage_store<F>(st, ihi, ilo, min(npad, CR));
  if (!dense_build_map(map, rp, cnt, map_bytes, MS, pre, g, n0, n, e0, ne)) return false;
  DFGNN_STAMP(1)
  if constexpr (NCHUNK == 1) dense_stage_load<F>(st, Vb, hf, 0, min(npad, CR), n);  // lands during the S phase

  f32x4 S[NT];
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    if (c > 0) {
      __syncthreads();
      dense_stage<F>(ihi, ilo, Kb, hf, c * CR, min(npad - c * CR, CR), n);
      __syncthreads();
    }
    if (active) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jt = 8 * c + u;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (jt < ntile) {
          const size_t off = (size_t)(16 * u + mi) * RS + 8 * mq;
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            const bf16x8 kh = *reinterpret_cast<const bf16x8 *>(ihi + off + 32 * t);
            const bf16x8 kl = *reinterpret_cast<const bf16x8 *>(ilo + off + 32 * t);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[t], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[t], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[t], acc, 0, 0, 0);
          }
        }
        S[jt] = acc;  // lane (mi, mq), register r: S[i = strip row mi][j = 16 jt + 4 mq + r]
      }
    }
  }
  DFGNN_STAMP(2)

  // ---- masked row softmax, in registers -------------------------------------------------------------------------
  float inv = 0.f;
  if (active) {
    const unsigned char *mrow = map + (size_t)i * MS + 4 * mq;
    float mx = -INFINITY;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const unsigned w = (jt < ntile) ? *reinterpret_cast<const unsigned *>(mrow + 16 * jt) : 0xFFFFFFFFu;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool edge = ((w >> (8 * r)) & 0xFFu) != 0xFFu;
        const float s = edge ? S[jt][r] : -INFINITY;
        S[jt][r] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = xor16_32_max(mx);
    const float base = (mx == -INFINITY) ? 0.f : mx;
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = fast_exp(S[jt][r] - base);  // exp(-inf) = 0 for the masked pairs
        S[jt][r] = p;
        sum += p;
      }
    sum = xor16_32_sum(sum);
    inv = (sum != 0.f) ? 1.f / sum : 0.f;
    if constexpr (WRITE_ATTN) {
      // attn_edge (CSR order): through LDS when the range's edge array fits (then each strip streams its own
      // contiguous slice out), else straight from the registers (scattered 4-byte stores)
      if (i < n) {
        float *lrow = pstage + rp[i];
        float *grow = attn_edge + (size_t)head * g.nnz + e0 + rp[i];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          if (jt < ntile) {
            const unsigned w = *reinterpret_cast<const unsigned *>(mrow + 16 * jt);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const unsigned slot = (w >> (8 * r)) & 0xFFu;
              if (slot != 0xFFu) {
                if (stage_attn) lrow[slot] = S[jt][r] * inv;
                else grow[slot] = S[jt][r] * inv;
              }
            }
          }
        }
      }
      if (stage_attn) {
        wave_sync();
        const int s0 = rp[wave * 16], s1 = rp[min(n, wave * 16 + 16)];
        float *dst = attn_edge + (size_t)head * g.nnz + e0;
        for (int e = s0 + lane; e < s1; e += kWave) dst[e] = pstage[e];
      }
    }
  }
  DFGNN_STAMP(3)

  // ---- O^T = V^T P^T ---------------------------------------------------------------------------------------------
  f32x4 o[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) o[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
  // transposed-read addressing: lane 4q + p of a 16-lane group points at row q, columns 4p .. 4p+3 of the block
  const int tq = mi >> 2, tp = mi & 3;
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c) {
    __syncthreads();  // every strip is done with the previous image
    if (NCHUNK == 1) dense_stage_store<F>(st, ihi, ilo, min(npad, CR));
    else dense_stage<F>(ihi, ilo, Vb, hf, c * CR, min(npad - c * CR, CR), n);
    __syncthreads();
    if (c == 0) { DFGNN_STAMP(4) }
    if (active) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jb = 4 * c + u;  // 32-column block of P
        if (2 * jb < ntile) {
          bf16x8 ph, pl;
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const float p = (t < 4) ? S[2 * jb][t] : S[2 * jb + 1][t - 4];
            const __bf16 h = (__bf16)p;
            ph[t] = h;
            pl[t] = (__bf16)(p - (float)h);
          }
          const size_t voff = (size_t)(32 * u + 4 * mq + tq) * RS + 4 * tp;
#pragma unroll
          for (int ft = 0; ft < FT; ++ft) {
            const bf16x8 vh = dense_tr_pair(ihi + voff + 16 * ft, 16 * RS);
            const bf16x8 vl = dense_tr_pair(ilo + voff + 16 * ft, 16 * RS);
            o[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph, o[ft], 0, 0, 0);
            o[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph, o[ft], 0, 0, 0);
            o[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl, o[ft], 0, 0, 0);
          }
        }
      }
    }
  }
  // lane (mi, mq), register r of tile ft: O[i][16 ft + 4 mq + r]
  if (active && i < n) {
    float *orow = out + (size_t)(n0 + i) * hf + (size_t)head * F + 4 * mq;
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
      *reinterpret_cast<float4 *>(orow + 16 * ft) =
          make_float4(o[ft][0] * inv, o[ft][1] * inv, o[ft][2] * inv, o[ft][3] * inv);
  }
  DFGNN_STAMP(5)
  if (threadIdx.x == 0) { DFGNN_STAMP(6) }
  return true;
}

}  // namespace dfgnn
