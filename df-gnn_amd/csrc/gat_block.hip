// gat_block.hip -- GAT 'hyper' forward for batched (block-diagonal) graphs: one workgroup per closed node
// range of the block plan, the range's feature rows X resident in LDS (plan.hip / gt_block.hip describe the
// scheme).  Replaces fused_gat_hyper_inference{,_vec4} (DFGNN/src/fused_gatconv/fused_gatconv_hyper.cu:5-224)
// for batches; the logits need no dot product, so there is a single pass:
//   s_e = LeakyReLU(attn_row[i] + attn_col[j])   (attn_col of the range staged in LDS)
//   softmax in registers (degree <= 64) or through the per-edge array, then  out_i = sum_e P_e X_j  from LDS.
#include "dfgnn_block.hpp"

namespace dfgnn {

template <class C>
__global__ __launch_bounds__(kBlockThreads) void gat_block_fwd_kernel(Csr g, const int *__restrict__ fit,
                                                                      const float *__restrict__ attn_row,
                                                                      const float *__restrict__ attn_col, float slope,
                                                                      const float *__restrict__ X,
                                                                      float *__restrict__ edge_ws,
                                                                      float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int G = C::G, F = C::G * C::VEC * C::NCH;
  const int n0 = fit[2 * blockIdx.x], n1raw = fit[2 * blockIdx.x + 1];
  const bool edge_global = (n1raw & kPlanEdgeGlobal) != 0;
  const int n1 = n1raw & kPlanRangeMask;
  const int n = n1 - n0;
  const int head = blockIdx.y, h = g.h;
  const size_t hf = (size_t)h * F;
  const int e0 = g.row_ptr[n0];
  const int ne = g.row_ptr[n1] - e0;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int gid = lane / G, gl = lane % G;
  const BlockLds L = carve_block_lds(lds, n, edge_global ? 0 : ne, F, wave);
  const EdgeArr W{edge_global ? nullptr : L.lw, edge_global ? edge_ws + (size_t)head * g.nnz + e0 : nullptr};
  const bool narrow = n <= 256;
  float *acol = L.rinv;  // [n] attn_col of the range's nodes (the 1/sum slot of the GT layout is free here)

  load_block_index(L, g, n0, n, e0, ne);
  for (int i = threadIdx.x; i < n; i += kBlockThreads) acol[i] = attn_col[(size_t)(n0 + i) * h + head];
  load_resident(L.res, X + (size_t)n0 * hf + (size_t)head * F, n, F, hf);
  __syncthreads();

  for (int r = wave; r < n; r += kBlockWaves) {
    const int lb = __builtin_amdgcn_readfirstlane(L.rp[r]);
    const int deg = __builtin_amdgcn_readfirstlane(L.rp[r + 1]) - lb;
    const float ar = attn_row[(size_t)(n0 + r) * h + head];
    float *out_row = out + (size_t)(n0 + r) * hf + (size_t)head * F;
    Frag<C> acc;
    frag_zero<C>(acc);
    float inv;
    if (deg <= kWave) {
      int c = 0;
      float s = -INFINITY;
      if (lane < deg) {
        c = block_col(L, narrow, lb + lane);
        s = leaky_relu(ar + acol[c], slope);
      }
      const float mx = wave_max(s);
      const float p = (s == -INFINITY) ? 0.f : fast_exp(s - mx);
      const float sum = wave_sum(p);
      inv = (sum != 0.f) ? 1.f / sum : 0.f;
      block_spmm<C>(acc, L.res, L.sc, deg, lane, gid, gl, [&](int, int &row, float &w) {
        row = c;
        w = p;
      });
    } else {
      float mx = -INFINITY;
      for (int e = lane; e < deg; e += kWave) {
        const float s = leaky_relu(ar + acol[block_col(L, narrow, lb + e)], slope);
        W.store(lb + e, s);
        mx = fmaxf(mx, s);
      }
      mx = wave_max(mx);
      float sum = 0.f;
      for (int e = lane; e < deg; e += kWave) {  // a lane re-reads only the slots it wrote itself
        const float p = fast_exp(W.load(lb + e) - mx);
        W.store(lb + e, p);
        sum += p;
      }
      sum = wave_sum(sum);
      inv = (sum != 0.f) ? 1.f / sum : 0.f;
      block_spmm<C>(acc, L.res, L.sc, deg, lane, gid, gl, [&](int k, int &row, float &w) {
        row = block_col(L, narrow, lb + k);
        w = W.load(lb + k);
      });
    }
    block_store_row<C>(acc, inv, out_row, gid, gl);
  }
}

int launch_gat_block_fwd(const Csr &g, const Plan &p, const float *attn_row, const float *attn_col, float slope,
                         const float *X, float *edge_ws, float *out, hipStream_t s) {
  if (p.num_fit == 0) return 0;
  // dense ranges (first in the list) go to the matrix-core kernel (gt_dense.hip: masked dense attention, P X on MFMA)
  const int first = (g.rows && dense_enabled()) ? p.num_dense : 0;
  if (first > 0)
    if (int rc = launch_gat_dense_fwd(g, p, attn_row, attn_col, slope, X, out, s)) return rc;
  if (first == p.num_fit) return 0;
  if (p.num_edge_global > 0 && !edge_ws) return kErrBadArg;
  const dim3 grid(p.num_fit - first, g.h);
  const int *fit = p.fit() + 2 * (size_t)first;
  const size_t lds = block_lds_bytes(p, g.f);
  return dispatch_vec4(g.f, [&](auto cfg) {
    using C = decltype(cfg);
    if (int rc = set_max_lds_cached(gat_block_fwd_kernel<C>)) return rc;
    gat_block_fwd_kernel<C><<<grid, kBlockThreads, lds, s>>>(g, fit, attn_row, attn_col, slope, X, edge_ws, out);
    return launch_status();
  });
}

}  // namespace dfgnn
