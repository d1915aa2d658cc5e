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
#include "dfgnn_block.hpp"

namespace dfgnn {

template <class C, bool WRITE_ATTN>
__global__ __launch_bounds__(kBlockThreads) void gt_block_fwd_kernel(Csr g, const int *__restrict__ fit,
                                                                     const float *__restrict__ Q,
                                                                     const float *__restrict__ K,
                                                                     const float *__restrict__ V,
                                                                     float *__restrict__ attn_edge,
                                                                     float *__restrict__ edge_ws,
                                                                     float *__restrict__ out, int lds_bytes) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int G = C::G, EPW = C::EPW, F = C::G * C::VEC * C::NCH;  // F == g.f (checked by the launcher)
  const int n0 = fit[2 * blockIdx.x], n1raw = fit[2 * blockIdx.x + 1];
  const bool edge_global = (n1raw & kPlanEdgeGlobal) != 0;
  const int n1 = n1raw & kPlanRangeMask;
  const int n = n1 - n0;
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * F;
  const int e0 = g.row_ptr[n0];
  const int ne = g.row_ptr[n1] - e0;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int gid = lane / G, gl = lane % G;
  const bool w_global = edge_global;
  const BlockLds L = carve_block_lds(lds, n, w_global ? 0 : ne, F, wave);
  // exp values between the passes: LDS, or (large ranges) this range's slice of the global scratch
  const EdgeArr W{w_global ? nullptr : L.lw, w_global ? edge_ws + (size_t)head * g.nnz + e0 : nullptr};
  int *sci = reinterpret_cast<int *>(L.sc);  // pass A stages columns only (first 256 B of the scratch)
  const float *Qh = Q + (size_t)head * F;
  float *outh = out + (size_t)head * F;
  float *attn_h = WRITE_ATTN ? attn_edge + (size_t)head * g.nnz : nullptr;
  const bool narrow = n <= 256;
  const int kk = chunk_edge_of_lane<C>(gid, gl);  // the edge of a chunk whose logit this lane keeps
  const int stage = (lane % EPW) * G + lane / EPW;  // where lane's own edge goes in the de-interleaved scratch

  DFGNN_STAMP(0)
  load_block_index(L, g, n0, n, e0, ne);
  load_resident(L.res, K + (size_t)n0 * hf + (size_t)head * F, n, F, hf);
  __syncthreads();
  DFGNN_STAMP(1)

  // ---- pass A: logits of every edge of the range, K served from LDS; then the row softmax ----------
  {
    Frag<C> q_next;
    if (wave < n) frag_load_full<C>(q_next, Qh + (size_t)(n0 + wave) * hf, gl);
    for (int r = wave; r < n; r += kBlockWaves) {
      // block-relative edge range of row n0 + r (wave-uniform: keep the loop control in SGPRs)
      const int lb = __builtin_amdgcn_readfirstlane(L.rp[r]);
      const int deg = __builtin_amdgcn_readfirstlane(L.rp[r + 1]) - lb;
      const Frag<C> q = q_next;
      if (r + kBlockWaves < n) frag_load_full<C>(q_next, Qh + (size_t)(n0 + r + kBlockWaves) * hf, gl);
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
        if (kk < deg) W.store(lb + kk, p);
        if (lane == 0) L.rinv[r] = (sum != 0.f) ? 1.f / sum : 0.f;
      } else {
        for (int c0 = 0; c0 < deg; c0 += kWave) {
          const int nt = min(kWave, deg - c0);
          const float myval = (g.val && kk < nt) ? g.val[e0 + lb + c0 + kk] : 1.f;
          sci[stage] = (lane < nt) ? block_col(L, narrow, lb + c0 + lane) : 0;
          wave_sync();
          const float mine = block_chunk_logits<C>(L.res, sci, q, nt, gid, gl);
          wave_sync();
          if (kk < nt) W.store(lb + c0 + kk, mine * myval);
        }
        wave_sync();
        // lane l wrote slot c0 + kk(l) and now reads slot l + 64 j: order the wave's global traffic first
        if (w_global) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        float mx = -INFINITY;
        for (int e = lane; e < deg; e += kWave) mx = fmaxf(mx, W.load(lb + e));
        mx = wave_max(mx);
        float sum = 0.f;
        for (int e = lane; e < deg; e += kWave) {
          const float s = W.load(lb + e);
          const float p = (s == -INFINITY) ? 0.f : fast_exp(s - mx);
          W.store(lb + e, p);
          sum += p;
        }
        sum = wave_sum(sum);
        if (lane == 0) L.rinv[r] = (sum != 0.f) ? 1.f / sum : 0.f;
      }
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
    Frag<C> acc;
    frag_zero<C>(acc);
    block_spmm<C>(acc, L.res, L.sc, deg, lane, gid, gl, [&](int k, int &row, float &w) {
      w = W.load(lb + k);
      row = block_col(L, narrow, lb + k);
      if constexpr (WRITE_ATTN) attn_h[e0 + lb + k] = w * inv;
    });
    block_store_row<C>(acc, inv, outh + (size_t)i * hf, gid, gl);
  }
  DFGNN_STAMP(5)
  if (threadIdx.x == 0) { DFGNN_STAMP(6) }
#ifdef DFGNN_STAMPS
  if (threadIdx.x == 0 && dfgnn_stamps) dfgnn_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + 7] = ((unsigned long long)n << 32) | (unsigned)ne;
#endif
}

size_t block_lds_bytes(const Plan &p, int f) {
  if (p.num_edge_global > 0) return kLdsBytes;
  const size_t n = p.max_fit_nodes, e = p.max_fit_edges;
  size_t b = n * f * 4 + (e + 4) * 4 + (n + 4) * 4 + (n + 8) * 4 + kBlockScratchBytes + (e + 4) * 2 + 64;
  return b > (size_t)kLdsBytes ? (size_t)kLdsBytes : b;
}

// The resident kernels are compiled for exact row widths (f == G*VEC*NCH: 16, 32, 64, 128, 256, 512, 1024).
bool block_width_ok(int f) {
  return dispatch_vec4(f, [&](auto cfg) {
    using C = decltype(cfg);
    return (int)(C::G * C::VEC * C::NCH == f);
  }) == 1;
}

int launch_gt_block_fwd(const Csr &g, const Plan &p, const float *Q, const float *K, const float *V,
                        float *attn_edge, float *edge_ws, float *out, hipStream_t s) {
  if (p.num_fit == 0) return 0;
  // Dense ranges with unit edge values go to the matrix-core kernel (gt_dense.hip); they come first in the list.
  const int first = (!g.val && g.rows && dense_enabled()) ? p.num_dense : 0;
  if (first > 0)
    if (int rc = launch_gt_dense_fwd(g, p, Q, K, V, attn_edge, out, s)) return rc;
  if (first == p.num_fit) return 0;
  // Ranges whose exp values do not fit LDS park them in global memory: the training forward lends its own
  // attn_edge output for that (it is overwritten with the normalised values in pass B), inference needs
  // the caller's scratch.
  if (!edge_ws) edge_ws = attn_edge;
  if (p.num_edge_global > 0 && !edge_ws) return kErrBadArg;
  const dim3 grid(p.num_fit - first, g.h);
  const int *fit = p.fit() + 2 * (size_t)first;
  const size_t lds = block_lds_bytes(p, g.f);
  return dispatch_vec4(g.f, [&](auto cfg) {
    using C = decltype(cfg);
    if (attn_edge) {
      if (int rc = set_max_lds_cached(gt_block_fwd_kernel<C, true>)) return rc;
      gt_block_fwd_kernel<C, true><<<grid, kBlockThreads, lds, s>>>(g, fit, Q, K, V, attn_edge, edge_ws, out,
                                                                     (int)lds);
    } else {
      if (int rc = set_max_lds_cached(gt_block_fwd_kernel<C, false>)) return rc;
      gt_block_fwd_kernel<C, false><<<grid, kBlockThreads, lds, s>>>(g, fit, Q, K, V, nullptr, edge_ws, out,
                                                                      (int)lds);
    }
    return launch_status();
  });
}

}  // namespace dfgnn

#ifdef DFGNN_STAMPS
extern "C" int dfgnn_debug_set_stamps(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_stamps), &p, sizeof(p));
}
extern "C" int dfgnn_debug_set_round_stamps(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_round_stamps), &p, sizeof(p));
}
#endif
