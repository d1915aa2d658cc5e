// gt_block_bwd.hip -- backward of the fused GT convolution for batched (block-diagonal) graphs:
// ONE kernel per closed node range (block plan, plan.hip), four passes, each with a different matrix of
// the range resident in LDS so that every per-edge 4f-byte row gather of the reference's two backward
// kernels (fused_gtconv_backward.cu:40-70 and :73-191, served by the L2 there) stays inside the CU:
//
//   pass 1  V resident   dP_e = <dO_i, V_j>;  t_i = sum_e P_e dP_e;  dS_e = P_e (dP_e - t_i)  -> LDS (x val_e)
//   pass 2  K resident   dQ_i = sum_e dS_e val_e K_j
//   pass 3  dO resident  dV_j = sum_{e -> j} P_e dO_i            (CSC walk, P gathered through val_idx)
//   pass 4  Q resident   dK_j = sum_{e -> j} dS_e val_e Q_i      (CSC walk, dS from LDS through val_idx)
//
// dS never goes to global memory (the reference materialises it as grad_edge); the range's CSC entries
// (row ids as bytes, CSR slots as 16-bit offsets) are staged in LDS once when they fit.
#include "dfgnn_block.hpp"

namespace dfgnn {

// Diagnostic build only (-DDFGNN_STAMPS): phase boundaries in shader cycles (see gt_block.hip).
#ifdef DFGNN_STAMPS
__device__ unsigned long long *dfgnn_bwd_stamps = nullptr;
#define DFGNN_BSTAMP(k)                                                                           \
  if (threadIdx.x == 0 && dfgnn_bwd_stamps)                                                       \
    dfgnn_bwd_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define DFGNN_BSTAMP(k)
#endif


struct BwdLds {
  float *res;              // [n * F]   resident rows: V, K, dO, Q in turn
  float *lw;               // [ne]      dP, then dS * val  (CSR order)
  int *rp;                 // [n + 1]   row_ptr, relative to the range's first edge
  int *cp;                 // [n + 1]   col_ptr, relative to the range's first CSC entry
  int2 *sc;                // per-wave staging
  unsigned char *cols;     // [ne] (or 2 ne)  block-local column ids
  unsigned char *ri;       // [ne]      block-local row id of each CSC entry      (only if staged)
  unsigned short *vi;      // [ne]      block-local CSR slot of each CSC entry    (only if staged)
  bool staged;
};

__device__ __forceinline__ BwdLds carve_bwd_lds(float *lds, int n, int ne, int ne_lds, int f, int wave) {
  BwdLds b;
  b.res = lds;
  b.lw = b.res + (size_t)n * f;
  b.rp = reinterpret_cast<int *>(b.lw + ((ne_lds + 3) & ~3));
  b.cp = b.rp + ((n + 1 + 3) & ~3);
  int2 *sc0 = reinterpret_cast<int2 *>(b.cp + ((n + 1 + 3) & ~3));
  b.sc = sc0 + wave * kWave;
  b.cols = reinterpret_cast<unsigned char *>(sc0 + kBlockWaves * kWave);
  const bool narrow = n <= 256;
  const size_t cols_bytes = ((size_t)ne * (narrow ? 1 : 2) + 7) & ~(size_t)3;
  b.ri = b.cols + cols_bytes;
  b.vi = reinterpret_cast<unsigned short *>(b.ri + (((size_t)ne + 7) & ~(size_t)3));
  const size_t end = (size_t)(reinterpret_cast<unsigned char *>(b.vi) - reinterpret_cast<unsigned char *>(lds)) +
                     (size_t)ne * 2 + 8;
  b.staged = narrow && ne < 65536 && end <= (size_t)kLdsBytes;
  return b;
}

template <class C>
__global__ __launch_bounds__(kBlockThreads) void gt_block_bwd_kernel(
    Csr g, const int *__restrict__ fit, const int *__restrict__ col_ptr, const int *__restrict__ row_ind,
    const int *__restrict__ val_idx, const float *__restrict__ Q, const float *__restrict__ K,
    const float *__restrict__ V, const float *__restrict__ attn_edge, const float *__restrict__ dO,
    float *__restrict__ edge_ws, float *__restrict__ dQ, float *__restrict__ dK, float *__restrict__ dV) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int G = C::G, EPW = C::EPW, F = C::G * C::VEC * C::NCH;
  const int n0 = fit[2 * blockIdx.x], n1raw = fit[2 * blockIdx.x + 1];
  const bool edge_global = (n1raw & kPlanEdgeGlobal) != 0;
  const int n1 = n1raw & kPlanRangeMask;
  const int n = n1 - n0;
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * F;
  const int e0 = g.row_ptr[n0];
  const int ne = g.row_ptr[n1] - e0;
  const int ce0 = col_ptr[n0];  // == e0 for a closed range; kept separate for clarity
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int gid = lane / G, gl = lane % G;
  const BwdLds L = carve_bwd_lds(lds, n, ne, edge_global ? 0 : ne, F, wave);
  // dP, then dS * val, per edge of the range: LDS, or (large ranges) the caller's grad_edge scratch
  const EdgeArr W{edge_global ? nullptr : L.lw, edge_global ? edge_ws + (size_t)head * g.nnz + e0 : nullptr};
  int *sci = reinterpret_cast<int *>(L.sc);
  const bool narrow = n <= 256;
  const int kk = chunk_edge_of_lane<C>(gid, gl);
  const int stage = (lane % EPW) * G + lane / EPW;
  const size_t hoff = (size_t)head * F;
  const float *P_h = attn_edge + (size_t)head * g.nnz + e0;  // this range's slice of attn_edge
  const float *valb = g.val ? g.val + e0 : nullptr;

  // ---- stage index arrays + V ------------------------------------------------------------------------------
  DFGNN_BSTAMP(0)
  for (int i = threadIdx.x; i <= n; i += kBlockThreads) {
    L.rp[i] = g.row_ptr[n0 + i] - e0;
    L.cp[i] = col_ptr[n0 + i] - ce0;
  }
  {
    const int *ci = g.col_ind + e0;
    if (narrow) {
      for (int b = threadIdx.x * 4; b < ne; b += kBlockThreads * 4) {
        unsigned v = 0, r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (b + k < ne) {
            v |= (unsigned)(ci[b + k] - n0) << (8 * k);
            if (L.staged) r |= (unsigned)(row_ind[ce0 + b + k] - n0) << (8 * k);
          }
        *reinterpret_cast<unsigned *>(L.cols + b) = v;
        if (L.staged) *reinterpret_cast<unsigned *>(L.ri + b) = r;
      }
      if (L.staged)
        for (int b = threadIdx.x * 2; b < ne; b += kBlockThreads * 2) {
          unsigned v = (unsigned)(val_idx[ce0 + b] - e0);
          if (b + 1 < ne) v |= (unsigned)(val_idx[ce0 + b + 1] - e0) << 16;
          *reinterpret_cast<unsigned *>(L.vi + b) = v;
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
  auto col_of = [&](int e) -> int {
    return narrow ? (int)L.cols[e] : (int)reinterpret_cast<const unsigned short *>(L.cols)[e];
  };
  load_resident(L.res, V + (size_t)n0 * hf + hoff, n, F, hf);
  __syncthreads();
  DFGNN_BSTAMP(1)

  // ---- pass 1 (V resident): dP, row sums, dS -> lw --------------------------------------------------------
  {
    const float *dOh = dO + hoff;
    Frag<C> g_next;
    if (wave < n) frag_load_full<C>(g_next, dOh + (size_t)(n0 + wave) * hf, gl);
    for (int r = wave; r < n; r += kBlockWaves) {
      const int lb = __builtin_amdgcn_readfirstlane(L.rp[r]);
      const int deg = __builtin_amdgcn_readfirstlane(L.rp[r + 1]) - lb;
      const Frag<C> go = g_next;
      if (r + kBlockWaves < n) frag_load_full<C>(g_next, dOh + (size_t)(n0 + r + kBlockWaves) * hf, gl);
      if (deg <= kWave) {
        const bool mine_ok = kk < deg;
        const float pk = mine_ok ? P_h[lb + kk] : 0.f;
        const float vk = (valb && mine_ok) ? valb[lb + kk] : 1.f;
        sci[stage] = (lane < deg) ? col_of(lb + lane) : 0;
        wave_sync();
        const float dp = block_chunk_logits<C>(L.res, sci, go, deg, gid, gl);
        wave_sync();
        const float t = wave_sum(mine_ok ? pk * dp : 0.f);
        if (mine_ok) W.store(lb + kk, pk * (dp - t) * vk);
      } else {
        float tacc = 0.f;
        for (int c0 = 0; c0 < deg; c0 += kWave) {
          const int nt = min(kWave, deg - c0);
          const float pk = (kk < nt) ? P_h[lb + c0 + kk] : 0.f;
          sci[stage] = (lane < nt) ? col_of(lb + c0 + lane) : 0;
          wave_sync();
          const float dp = block_chunk_logits<C>(L.res, sci, go, nt, gid, gl);
          wave_sync();
          if (kk < nt) {
            W.store(lb + c0 + kk, dp);
            tacc = fmaf(pk, dp, tacc);
          }
        }
        const float t = wave_sum(tacc);
        wave_sync();
        if (edge_global) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        for (int e = lane; e < deg; e += kWave) {
          const float ds = P_h[lb + e] * (W.load(lb + e) - t);
          W.store(lb + e, valb ? ds * valb[lb + e] : ds);
        }
      }
    }
  }
  DFGNN_BSTAMP(2)
  __syncthreads();
  DFGNN_BSTAMP(3)

  // ---- pass 2 (K resident): dQ ---------------------------------------------------------------------------------
  load_resident(L.res, K + (size_t)n0 * hf + hoff, n, F, hf);
  __syncthreads();
  DFGNN_BSTAMP(4)
  for (int r = wave; r < n; r += kBlockWaves) {
    const int lb = __builtin_amdgcn_readfirstlane(L.rp[r]);
    const int deg = __builtin_amdgcn_readfirstlane(L.rp[r + 1]) - lb;
    Frag<C> acc;
    frag_zero<C>(acc);
    block_spmm<C>(acc, L.res, L.sc, deg, lane, gid, gl, [&](int k, int &row, float &w) {
      w = W.load(lb + k);
      row = col_of(lb + k);
    });
    block_store_row<C>(acc, 1.f, dQ + (size_t)(n0 + r) * hf + hoff, gid, gl);
  }
  DFGNN_BSTAMP(5)
  __syncthreads();
  DFGNN_BSTAMP(6)

  // CSC entry t of this range -> (block-local row id, block-local CSR slot)
  auto csc_entry = [&](int t, int &row, int &slot) {
    if (L.staged) {
      row = L.ri[t];
      slot = L.vi[t];
    } else {
      row = row_ind[ce0 + t] - n0;
      slot = val_idx[ce0 + t] - e0;
    }
  };

  // ---- pass 3 (dO resident): dV --------------------------------------------------------------------------------
  load_resident(L.res, dO + (size_t)n0 * hf + hoff, n, F, hf);
  __syncthreads();
  DFGNN_BSTAMP(7)
  for (int c = wave; c < n; c += kBlockWaves) {
    const int cb = __builtin_amdgcn_readfirstlane(L.cp[c]);
    const int cdeg = __builtin_amdgcn_readfirstlane(L.cp[c + 1]) - cb;
    Frag<C> acc;
    frag_zero<C>(acc);
    block_spmm<C>(acc, L.res, L.sc, cdeg, lane, gid, gl, [&](int k, int &row, float &w) {
      int slot;
      csc_entry(cb + k, row, slot);
      w = P_h[slot];
    });
    block_store_row<C>(acc, 1.f, dV + (size_t)(n0 + c) * hf + hoff, gid, gl);
  }
  DFGNN_BSTAMP(8)
  __syncthreads();
  DFGNN_BSTAMP(9)

  // ---- pass 4 (Q resident): dK ---------------------------------------------------------------------------------
  load_resident(L.res, Q + (size_t)n0 * hf + hoff, n, F, hf);
  __syncthreads();
  DFGNN_BSTAMP(10)
  for (int c = wave; c < n; c += kBlockWaves) {
    const int cb = __builtin_amdgcn_readfirstlane(L.cp[c]);
    const int cdeg = __builtin_amdgcn_readfirstlane(L.cp[c + 1]) - cb;
    Frag<C> acc;
    frag_zero<C>(acc);
    block_spmm<C>(acc, L.res, L.sc, cdeg, lane, gid, gl, [&](int k, int &row, float &w) {
      int slot;
      csc_entry(cb + k, row, slot);
      w = W.load(slot);
    });
    block_store_row<C>(acc, 1.f, dK + (size_t)(n0 + c) * hf + hoff, gid, gl);
  }
  DFGNN_BSTAMP(11)
}

int launch_gt_block_bwd(const Csr &g, const Plan &p, const int *col_ptr, const int *row_ind, const int *val_idx,
                        const float *Q, const float *K, const float *V, const float *attn_edge,
                        const float *grad_out, float *edge_ws, float *dQ, float *dK, float *dV, hipStream_t s) {
  if (p.num_fit == 0) return 0;
  const int first = (!g.val && g.rows && dense_enabled()) ? p.num_dense : 0;  // matrix-core kernel, gt_dense.hip
  if (first > 0)
    if (int rc = launch_gt_dense_bwd(g, p, Q, K, V, attn_edge, grad_out, dQ, dK, dV, s)) return rc;
  if (first == p.num_fit) return 0;
  if (p.num_edge_global > 0 && !edge_ws) return kErrBadArg;
  const dim3 grid(p.num_fit - first, g.h);
  const int *fit = p.fit() + 2 * (size_t)first;
  return dispatch_vec4(g.f, [&](auto cfg) {
    using C = decltype(cfg);
    if (int rc = set_max_lds_cached(gt_block_bwd_kernel<C>)) return rc;
    gt_block_bwd_kernel<C><<<grid, kBlockThreads, kLdsBytes, s>>>(g, fit, col_ptr, row_ind, val_idx, Q, K, V,
                                                                  attn_edge, grad_out, edge_ws, dQ, dK, dV);
    return launch_status();
  });
}

}  // namespace dfgnn

#ifdef DFGNN_STAMPS
extern "C" int dfgnn_debug_set_bwd_stamps(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_bwd_stamps), &p, sizeof(p));
}
#endif
