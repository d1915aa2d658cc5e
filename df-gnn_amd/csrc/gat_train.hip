// gat_train.hip -- GAT training pair for gfx950, general kernels (any graph, no degree limit) and their lane-group
// forms for low-degree graphs (gat_rowgroup_*: nnz < 8 m, as for the GT path).
//
//   gat_train_fwd_kernel   CSR, a wave per row, 64-edge tiles, online softmax; also writes the row statistics the
//                          backward recomputes P from (edge_max, edge_sum) and applies attention dropout from a
//                          caller-provided tensor of uniform randoms.  replaces fused_forward_kernel
//                          (DFGNN/src/fused_gatconv/fused_gatconv_kernel.cu:24-125; launcher :1062-1129)
//   gat_bwd_rows_kernel    CSR pass: dP_e = <dO[i], X[j]>, g_e = keep_e dP_e / (1 - drop),
//                          G_e = P_e (g_e - sum_row P g) LeakyReLU'(pre_e) -> grad_edge; grad_attn_row[i] = sum_e G_e.
//                          replaces mhsddmm + fused_backward_kernel (fused_gatconv_kernel.cu:711-865)
//   gat_bwd_cols_kernel    CSC pass: grad_feat[j] = sum_{e->j} keep_e P_e / (1 - drop) dO[i],
//                          grad_attn_col[j] = sum_{e->j} G_e.  replaces mhspmm_backward_kernel (:609-660) and the
//                          atomicAdd into grad_attn_col (:853): the column sums are deterministic here.
//
// Layouts as in the reference: edge_max / edge_sum / attn_row / attn_col fp32[m, h]; edge_mask fp32[nnz, h]
// (EDGE-major, CSR order, fused_gatconv_kernel.cu:101).  grad_edge is this library's scratch, fp32[h, nnz].
#include <cstdlib>

#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

// d_e = <a, X[cols[e]]> for the nt (<= 64) edges of a tile; lane 0 of each group writes sw[e].  4 gathers in flight.
template <class C>
__device__ __forceinline__ void tile_dots(const Frag<C> &a, const int *cols, int nt, const float *__restrict__ X,
                                          size_t hf, int f, int gid, int gl, float *sw) {
  int e = gid;
  for (; e + 3 * C::EPW < nt; e += 4 * C::EPW) {
    Frag<C> x0, x1, x2, x3;
    frag_load<C>(x0, X + (size_t)cols[e] * hf, f, gl);
    frag_load<C>(x1, X + (size_t)cols[e + C::EPW] * hf, f, gl);
    frag_load<C>(x2, X + (size_t)cols[e + 2 * C::EPW] * hf, f, gl);
    frag_load<C>(x3, X + (size_t)cols[e + 3 * C::EPW] * hf, f, gl);
    const float d0 = lanes_sum<C::G>(frag_dot<C>(a, x0)), d1 = lanes_sum<C::G>(frag_dot<C>(a, x1));
    const float d2 = lanes_sum<C::G>(frag_dot<C>(a, x2)), d3 = lanes_sum<C::G>(frag_dot<C>(a, x3));
    if (gl == 0) {
      sw[e] = d0;
      sw[e + C::EPW] = d1;
      sw[e + 2 * C::EPW] = d2;
      sw[e + 3 * C::EPW] = d3;
    }
  }
  for (; e < nt; e += C::EPW) {
    Frag<C> x0;
    frag_load<C>(x0, X + (size_t)cols[e] * hf, f, gl);
    const float d0 = lanes_sum<C::G>(frag_dot<C>(a, x0));
    if (gl == 0) sw[e] = d0;
  }
}

// Rows (CSR pass) / columns (CSC pass) a launch covers: all of them, grid-strided -- or, next to the matrix-core
// kernels on a batch with a block plan, only the closed ranges those kernels do not serve: two lists of (n0, n1) pairs
// (the plan's non-dense fit ranges, whose n1 carries flag bits, and its spill chunks), one workgroup per pair.
struct RowLists {
  const int *a, *b;
  int na, nb;
};
__device__ __forceinline__ void row_span(const RowLists &rl, int m, int wave, int &beg, int &end, int &step) {
  if (rl.na + rl.nb == 0) {
    beg = blockIdx.x * kWavesPerBlock + wave;
    end = m;
    step = gridDim.x * kWavesPerBlock;
  } else {
    const int *e = (int)blockIdx.x < rl.na ? rl.a + 2 * blockIdx.x : rl.b + 2 * (blockIdx.x - rl.na);
    beg = e[0] + wave;
    end = e[1] & kPlanRangeMask;
    step = kWavesPerBlock;
  }
}

struct GatDrop {         // attention dropout: keep edge e of head hd iff mask[e*h + hd] > drop
  const float *mask;     // uniform randoms [nnz, h]; NULL = keep everything
  float drop, scale;     // scale = 1 / (1 - drop)  (1 when mask == NULL)
};

template <class C>
__global__ __launch_bounds__(kBlock) void gat_train_fwd_kernel(Csr g, const float *__restrict__ attn_row,
                                                               const float *__restrict__ attn_col, float slope,
                                                               const float *__restrict__ X, GatDrop dr,
                                                               float *__restrict__ edge_max,
                                                               float *__restrict__ edge_sum,
                                                               float *__restrict__ out, RowLists rl) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  const int head = blockIdx.y, h = g.h, f = g.f;
  const size_t hf = (size_t)h * f;
  const float *Xh = X + (size_t)head * f;
  const float *acol_h = attn_col + head;
  const int gid = lane / C::G, gl = lane % C::G;
  int rbeg, rend, rstep;
  row_span(rl, g.m, wave, rbeg, rend, rstep);
  for (int r = rbeg; r < rend; r += rstep) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    const float ar = attn_row[(size_t)r * h + head];
    Frag<C> acc;
    frag_zero<C>(acc);
    float m_run = -INFINITY, l_run = 0.f;
    for (int t0 = 0; t0 < deg; t0 += kWave) {
      const int nt = min(kWave, deg - t0);
      float s = -INFINITY;
      int c = 0;
      bool keep = true;
      if (lane < nt) {
        c = g.col_ind[lb + t0 + lane];
        s = leaky_relu(ar + acol_h[(size_t)c * h], slope);
        if (dr.mask) keep = dr.mask[(size_t)(lb + t0 + lane) * h + head] > dr.drop;
      }
      sc[lane] = c;
      online_step<C>(s, lane, sw, acc, m_run, l_run);  // the row sum counts every edge, dropped or not
      if (!keep) sw[lane] = 0.f;
      wave_sync();
      spmm_accum<C>(acc, sw, sc, nt, Xh, hf, f, gid, gl);
      wave_sync();
    }
    const float inv = (l_run != 0.f) ? dr.scale / l_run : 0.f;
    frag_reduce_groups<C>(acc);
    if (gid == 0) frag_store_scaled<C>(acc, inv, out + (size_t)r * hf + (size_t)head * f, f, gl);
    if (lane == 0) {
      edge_max[(size_t)r * h + head] = deg > 0 ? m_run : -1e38f;  // the reference's sentinel (:46, :66)
      edge_sum[(size_t)r * h + head] = l_run;
    }
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gat_bwd_rows_kernel(Csr g, const float *__restrict__ attn_row,
                                                              const float *__restrict__ attn_col, float slope,
                                                              const float *__restrict__ X,
                                                              const float *__restrict__ edge_max,
                                                              const float *__restrict__ edge_sum, GatDrop dr,
                                                              const float *__restrict__ dO,
                                                              float *__restrict__ grad_edge,
                                                              float *__restrict__ grad_row, RowLists rl) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  const int head = blockIdx.y, h = g.h, f = g.f;
  const size_t hf = (size_t)h * f;
  const float *Xh = X + (size_t)head * f, *dOh = dO + (size_t)head * f;
  const float *acol_h = attn_col + head;
  float *G_h = grad_edge + (size_t)head * g.nnz;
  const int gid = lane / C::G, gl = lane % C::G;
  int rbeg, rend, rstep;
  row_span(rl, g.m, wave, rbeg, rend, rstep);
  for (int r = rbeg; r < rend; r += rstep) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    float rs = 0.f;
    if (deg > 0) {
      const float ar = attn_row[(size_t)r * h + head];
      const float mx = edge_max[(size_t)r * h + head], inv = 1.f / edge_sum[(size_t)r * h + head];
      Frag<C> go;
      frag_load<C>(go, dOh + (size_t)r * hf, f, gl);
      // sweep 1: g_e and t = sum_e P_e g_e.  Single-tile rows keep (P, g, slope factor) in registers; longer rows
      // park g_e in grad_edge and recompute P in sweep 2.
      float t = 0.f, p_keep = 0.f, g_keep = 0.f, lr_keep = 0.f;
      for (int t0 = 0; t0 < deg; t0 += kWave) {
        const int nt = min(kWave, deg - t0);
        int c = 0;
        if (lane < nt) c = g.col_ind[lb + t0 + lane];
        sc[lane] = c;
        wave_sync();
        tile_dots<C>(go, sc, nt, Xh, hf, f, gid, gl, sw);
        wave_sync();
        if (lane < nt) {
          const int e = lb + t0 + lane;
          const float pre = ar + acol_h[(size_t)c * h];
          const float p = fast_exp(leaky_relu(pre, slope) - mx) * inv;
          const bool keep = dr.mask ? dr.mask[(size_t)e * h + head] > dr.drop : true;
          const float ge = keep ? sw[lane] * dr.scale : 0.f;
          t = fmaf(p, ge, t);
          p_keep = p;
          g_keep = ge;
          lr_keep = pre > 0.f ? 1.f : slope;
          if (deg > kWave) G_h[e] = ge;
        }
        wave_sync();
      }
      t = lanes_sum<kWave>(t);
      if (deg <= kWave) {
        float ge = 0.f;
        if (lane < deg) {
          ge = p_keep * (g_keep - t) * lr_keep;
          G_h[lb + lane] = ge;
        }
        rs = lanes_sum<kWave>(ge);
      } else {
        for (int e = lb + lane; e < lb + deg; e += kWave) {
          const float pre = ar + acol_h[(size_t)g.col_ind[e] * h];
          const float p = fast_exp(leaky_relu(pre, slope) - mx) * inv;
          const float ge = p * (G_h[e] - t) * (pre > 0.f ? 1.f : slope);  // this lane parked G_h[e] in sweep 1
          G_h[e] = ge;
          rs += ge;
        }
        rs = lanes_sum<kWave>(rs);
      }
    }
    if (lane == 0) grad_row[(size_t)r * h + head] = rs;
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gat_bwd_cols_kernel(Csr g, const int *__restrict__ col_ptr,
                                                              const int *__restrict__ row_ind,
                                                              const int *__restrict__ permute,
                                                              const float *__restrict__ attn_row,
                                                              const float *__restrict__ attn_col, float slope,
                                                              const float *__restrict__ edge_max,
                                                              const float *__restrict__ edge_sum, GatDrop dr,
                                                              const float *__restrict__ grad_edge,
                                                              const float *__restrict__ dO,
                                                              float *__restrict__ grad_feat,
                                                              float *__restrict__ grad_col, RowLists rl) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  const int head = blockIdx.y, h = g.h, f = g.f;
  const size_t hf = (size_t)h * f;
  const float *dOh = dO + (size_t)head * f;
  const float *arow_h = attn_row + head, *mx_h = edge_max + head, *sum_h = edge_sum + head;
  const float *G_h = grad_edge + (size_t)head * g.nnz;
  const int gid = lane / C::G, gl = lane % C::G;
  int jbeg, jend, jstep;
  row_span(rl, g.m, wave, jbeg, jend, jstep);  // (closed ranges: a range's columns are its rows)
  for (int j = jbeg; j < jend; j += jstep) {
    const int lb = col_ptr[j], n = col_ptr[j + 1] - lb;
    const float ac = attn_col[(size_t)j * h + head];
    Frag<C> acc;
    frag_zero<C>(acc);
    float gs = 0.f;
    for (int t0 = 0; t0 < n; t0 += kWave) {
      const int nt = min(kWave, n - t0);
      float w = 0.f;
      int i = 0;
      if (lane < nt) {
        i = row_ind[lb + t0 + lane];
        const int e = permute[lb + t0 + lane];
        const float pre = arow_h[(size_t)i * h] + ac;
        const float p = fast_exp(leaky_relu(pre, slope) - mx_h[(size_t)i * h]) / sum_h[(size_t)i * h];
        const bool keep = dr.mask ? dr.mask[(size_t)e * h + head] > dr.drop : true;
        w = keep ? p * dr.scale : 0.f;
        gs += G_h[e];
      }
      sw[lane] = w;
      sc[lane] = i;
      wave_sync();
      spmm_accum<C>(acc, sw, sc, nt, dOh, hf, f, gid, gl);
      wave_sync();
    }
    frag_reduce_groups<C>(acc);
    if (gid == 0) frag_store_scaled<C>(acc, 1.f, grad_feat + (size_t)j * hf + (size_t)head * f, f, gl);
    gs = lanes_sum<kWave>(gs);
    if (lane == 0) grad_col[(size_t)j * h + head] = gs;
  }
}

// ---- low-degree graphs (molecules, peptides: ~2 edges per row) -----------------------------------------------------
// A wave per row wastes most of its lanes there; as in gt_lowdeg.hip every group of G lanes (one feature row wide)
// owns one (row, head): EPW rows per wave, everything in registers, loops to each group's own degree.  The logits
// are scalars, so the row statistics are exact two-sweep values (no online rescaling).
template <class C>
__global__ __launch_bounds__(kBlock) void gat_rowgroup_fwd_kernel(Csr g, const float *__restrict__ attn_row,
                                                                  const float *__restrict__ attn_col, float slope,
                                                                  const float *__restrict__ X, GatDrop dr,
                                                                  float *__restrict__ edge_max,
                                                                  float *__restrict__ edge_sum,
                                                                  float *__restrict__ out) {
  constexpr int G = C::G;
  const int head = blockIdx.y, h = g.h, f = g.f;
  const size_t hf = (size_t)h * f, hoff = (size_t)head * f;
  const int gl = threadIdx.x % G;
  const int ngroups = gridDim.x * (kBlock / G);
  for (int r = blockIdx.x * (kBlock / G) + threadIdx.x / G; r < g.m; r += ngroups) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    const float ar = attn_row[(size_t)r * h + head];
    float mx = -INFINITY;
    for (int e = 0; e < deg; ++e)
      mx = fmaxf(mx, leaky_relu(ar + attn_col[(size_t)g.col_ind[lb + e] * h + head], slope));
    Frag<C> acc;
    frag_zero<C>(acc);
    float sum = 0.f;
    for (int e = 0; e < deg; ++e) {
      const int c = g.col_ind[lb + e];
      const float p = fast_exp(leaky_relu(ar + attn_col[(size_t)c * h + head], slope) - mx);
      sum += p;  // the row sum counts every edge, dropped or not
      const bool keep = dr.mask ? dr.mask[(size_t)(lb + e) * h + head] > dr.drop : true;
      Frag<C> x;
      frag_load<C>(x, X + (size_t)c * hf + hoff, f, gl);
      frag_fma<C>(acc, keep ? p : 0.f, x);
    }
    frag_store_scaled<C>(acc, sum != 0.f ? dr.scale / sum : 0.f, out + (size_t)r * hf + hoff, f, gl);
    if (gl == 0) {
      edge_max[(size_t)r * h + head] = deg > 0 ? mx : -1e38f;
      edge_sum[(size_t)r * h + head] = sum;
    }
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gat_rowgroup_bwd_rows_kernel(Csr g, const float *__restrict__ attn_row,
                                                                       const float *__restrict__ attn_col, float slope,
                                                                       const float *__restrict__ X,
                                                                       const float *__restrict__ edge_max,
                                                                       const float *__restrict__ edge_sum, GatDrop dr,
                                                                       const float *__restrict__ dO,
                                                                       float *__restrict__ grad_edge,
                                                                       float *__restrict__ grad_row) {
  constexpr int G = C::G;
  const int head = blockIdx.y, h = g.h, f = g.f;
  const size_t hf = (size_t)h * f, hoff = (size_t)head * f;
  const int gl = threadIdx.x % G;
  const int ngroups = gridDim.x * (kBlock / G);
  float *G_h = grad_edge + (size_t)head * g.nnz;
  for (int r = blockIdx.x * (kBlock / G) + threadIdx.x / G; r < g.m; r += ngroups) {
    const int lb = g.row_ptr[r], deg = g.row_ptr[r + 1] - lb;
    float rs = 0.f;
    if (deg > 0) {
      const float ar = attn_row[(size_t)r * h + head];
      const float mx = edge_max[(size_t)r * h + head], inv = 1.f / edge_sum[(size_t)r * h + head];
      Frag<C> go;
      frag_load<C>(go, dO + (size_t)r * hf + hoff, f, gl);
      float t = 0.f;
      for (int e = 0; e < deg; ++e) {  // g_e parked in grad_edge (every lane of the group holds the same value)
        const int c = g.col_ind[lb + e];
        Frag<C> x;
        frag_load<C>(x, X + (size_t)c * hf + hoff, f, gl);
        const float dp = lanes_sum<G>(frag_dot<C>(go, x));
        const bool keep = dr.mask ? dr.mask[(size_t)(lb + e) * h + head] > dr.drop : true;
        const float ge = keep ? dp * dr.scale : 0.f;
        const float p = fast_exp(leaky_relu(ar + attn_col[(size_t)c * h + head], slope) - mx) * inv;
        t = fmaf(p, ge, t);
        if (gl == 0) G_h[lb + e] = ge;
      }
      for (int e = 0; e < deg; ++e) {
        const float pre = ar + attn_col[(size_t)g.col_ind[lb + e] * h + head];
        const float p = fast_exp(leaky_relu(pre, slope) - mx) * inv;
        // lane 0 re-reads what it parked; the other lanes only need rs, which lane 0 writes
        const float ge = (gl == 0) ? p * (G_h[lb + e] - t) * (pre > 0.f ? 1.f : slope) : 0.f;
        if (gl == 0) G_h[lb + e] = ge;
        rs += ge;
      }
    }
    if (gl == 0) grad_row[(size_t)r * h + head] = rs;
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gat_rowgroup_bwd_cols_kernel(
    Csr g, const int *__restrict__ col_ptr, const int *__restrict__ row_ind, const int *__restrict__ permute,
    const float *__restrict__ attn_row, const float *__restrict__ attn_col, float slope,
    const float *__restrict__ edge_max, const float *__restrict__ edge_sum, GatDrop dr,
    const float *__restrict__ grad_edge, const float *__restrict__ dO, float *__restrict__ grad_feat,
    float *__restrict__ grad_col) {
  constexpr int G = C::G;
  const int head = blockIdx.y, h = g.h, f = g.f;
  const size_t hf = (size_t)h * f, hoff = (size_t)head * f;
  const int gl = threadIdx.x % G;
  const int ngroups = gridDim.x * (kBlock / G);
  const float *G_h = grad_edge + (size_t)head * g.nnz;
  for (int j = blockIdx.x * (kBlock / G) + threadIdx.x / G; j < g.m; j += ngroups) {
    const int lb = col_ptr[j], n = col_ptr[j + 1] - lb;
    const float ac = attn_col[(size_t)j * h + head];
    Frag<C> acc;
    frag_zero<C>(acc);
    float gs = 0.f;
    for (int t = 0; t < n; ++t) {
      const int i = row_ind[lb + t], e = permute[lb + t];
      const float p = fast_exp(leaky_relu(attn_row[(size_t)i * h + head] + ac, slope) - edge_max[(size_t)i * h + head]) /
                      edge_sum[(size_t)i * h + head];
      const bool keep = dr.mask ? dr.mask[(size_t)e * h + head] > dr.drop : true;
      Frag<C> go;
      frag_load<C>(go, dO + (size_t)i * hf + hoff, f, gl);
      frag_fma<C>(acc, keep ? p * dr.scale : 0.f, go);
      gs += G_h[e];
    }
    frag_store_scaled<C>(acc, 1.f, grad_feat + (size_t)j * hf + hoff, f, gl);
    if (gl == 0) grad_col[(size_t)j * h + head] = gs;
  }
}

// DFGNN_LOWDEG=0 in the environment (diagnostic switch, read once) keeps low-degree graphs on the wave-per-row kernels
static bool lowdeg_enabled() {
  static const bool on = [] { const char *e = getenv("DFGNN_LOWDEG"); return !e || atoi(e) != 0; }();
  return on;
}
static inline bool use_rowgroup(const Csr &g, const Plan *rest) {
  return !rest && lowdeg_enabled() && low_degree(g.m, g.nnz);
}

static dim3 rowgroup_grid(const Csr &g, int G) {
  const long per = kBlock / G;
  long blocks = ((long)g.m + per - 1) / per;
  if (blocks > 16384) blocks = 16384;
  return dim3((unsigned)(blocks < 1 ? 1 : blocks), g.h);
}

static inline int row_grid(int m, const RowLists &rl) {
  if (rl.na + rl.nb > 0) return rl.na + rl.nb;
  const long want = ((long)m + kWavesPerBlock - 1) / kWavesPerBlock;
  return (int)(want > (1 << 20) ? (1 << 20) : want);
}
// the ranges of a plan the matrix-core kernels do not serve (none: the whole graph)
static inline RowLists rest_of(const Plan *p) {
  if (!p) return RowLists{nullptr, nullptr, 0, 0};
  return RowLists{p->fit() + 2 * (size_t)p->num_dense, p->spill(), p->num_fit - p->num_dense, p->num_spill};
}

int launch_gat_train_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                         const float *edge_mask, float attn_drop, float *edge_max, float *edge_sum, float *out,
                         hipStream_t s, const Plan *rest) {
  const RowLists rl = rest_of(rest);
  const dim3 grid(row_grid(g.m, rl), g.h);
  const GatDrop dr{edge_mask, attn_drop, edge_mask ? 1.f / (1.f - attn_drop) : 1.f};
  const bool v4 = (g.f % 4 == 0) && aligned16(X) && aligned16(out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    if (use_rowgroup(g, rest))
      gat_rowgroup_fwd_kernel<C><<<rowgroup_grid(g, C::G), kBlock, 0, s>>>(g, attn_row, attn_col, slope, X, dr, edge_max,
                                                                          edge_sum, out);
    else
      gat_train_fwd_kernel<C><<<grid, kBlock, 0, s>>>(g, attn_row, attn_col, slope, X, dr, edge_max, edge_sum, out, rl);
    return launch_status();
  });
}

int launch_gat_bwd_rows(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                        const float *edge_max, const float *edge_sum, const float *edge_mask, float attn_drop,
                        const float *grad_out, float *grad_edge, float *grad_row, hipStream_t s, const Plan *rest) {
  const RowLists rl = rest_of(rest);
  const dim3 grid(row_grid(g.m, rl), g.h);
  const GatDrop dr{edge_mask, attn_drop, edge_mask ? 1.f / (1.f - attn_drop) : 1.f};
  const bool v4 = (g.f % 4 == 0) && aligned16(X) && aligned16(grad_out);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    if (use_rowgroup(g, rest))
      gat_rowgroup_bwd_rows_kernel<C><<<rowgroup_grid(g, C::G), kBlock, 0, s>>>(g, attn_row, attn_col, slope, X, edge_max,
                                                                               edge_sum, dr, grad_out, grad_edge, grad_row);
    else
      gat_bwd_rows_kernel<C><<<grid, kBlock, 0, s>>>(g, attn_row, attn_col, slope, X, edge_max, edge_sum, dr, grad_out,
                                                     grad_edge, grad_row, rl);
    return launch_status();
  });
}

int launch_gat_bwd_cols(const Csr &g, const int *col_ptr, const int *row_ind, const int *permute,
                        const float *attn_row, const float *attn_col, float slope, const float *edge_max,
                        const float *edge_sum, const float *edge_mask, float attn_drop, const float *grad_edge,
                        const float *grad_out, float *grad_feat, float *grad_col, hipStream_t s, const Plan *rest) {
  const RowLists rl = rest_of(rest);
  const dim3 grid(row_grid(g.m, rl), g.h);
  const GatDrop dr{edge_mask, attn_drop, edge_mask ? 1.f / (1.f - attn_drop) : 1.f};
  const bool v4 = (g.f % 4 == 0) && aligned16(grad_out) && aligned16(grad_feat);
  return dispatch_cfg(g.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    if (use_rowgroup(g, rest))
      gat_rowgroup_bwd_cols_kernel<C><<<rowgroup_grid(g, C::G), kBlock, 0, s>>>(
          g, col_ptr, row_ind, permute, attn_row, attn_col, slope, edge_max, edge_sum, dr, grad_edge, grad_out, grad_feat,
          grad_col);
    else
      gat_bwd_cols_kernel<C><<<grid, kBlock, 0, s>>>(g, col_ptr, row_ind, permute, attn_row, attn_col, slope, edge_max,
                                                     edge_sum, dr, grad_edge, grad_out, grad_feat, grad_col, rl);
    return launch_status();
  });
}

}  // namespace dfgnn
