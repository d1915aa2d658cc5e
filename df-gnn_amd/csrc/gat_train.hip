// gat_train.hip -- GAT training pair for gfx950: general kernels (any graph, no degree limit).
//
// Three passes, each as a per-row routine in two forms -- a wave per row (64-edge tiles) and, for low-degree graphs
// (nnz < 8 m, as on the GT path), a group of G lanes per row:
//   forward           CSR, online softmax; also writes the row statistics the backward recomputes P from
//                     (edge_max, edge_sum) and applies attention dropout from a caller-provided tensor of uniform
//                     randoms.  replaces fused_forward_kernel (DFGNN/src/fused_gatconv/fused_gatconv_kernel.cu:24-125;
//                     launcher :1062-1129)
//   backward, CSR     dP_e = <dO[i], X[j]>, g_e = keep_e dP_e / (1 - drop), G_e = P_e (g_e - sum_row P g)
//                     LeakyReLU'(pre_e) -> grad_edge; grad_attn_row[i] = sum_e G_e.  replaces mhsddmm +
//                     fused_backward_kernel (fused_gatconv_kernel.cu:711-865)
//   backward, CSC     grad_feat[j] = sum_{e->j} keep_e P_e / (1 - drop) dO[i], grad_attn_col[j] = sum_{e->j} G_e.
//                     replaces mhspmm_backward_kernel (:609-660) and the atomicAdd into grad_attn_col (:853): the
//                     column sums are deterministic here.
// gat_train_wave_kernel runs the wave form over the whole graph or over the ranges a block plan leaves to it;
// gat_train_group_kernel runs the lane-group form; a wave whose rows include a long one puts all its groups on each row.
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

// Everything the per-row routines need (pointers already offset to the head where that is a plain offset).
struct GatTrain {
  int m, nnz, h, f, head;
  size_t hf;
  const int *row_ptr, *col_ind;                 // CSR
  const int *col_ptr, *row_ind, *permute;       // CSC (column pass)
  const float *attn_row, *attn_col;             // [m, h]
  float slope;
  const float *Xh, *dOh;                        // features / output gradient, + head * f
  float *edge_max, *edge_sum;                   // [m, h]  (written by the forward, read by the backward)
  GatDrop dr;
  float *outh, *gfeath;                         // out / grad_feat, + head * f
  float *G_h;                                   // grad_edge + head * nnz
  float *grad_row, *grad_col;                   // [m, h]
  __device__ __forceinline__ size_t nh(int node) const { return (size_t)node * h + head; }
  __device__ __forceinline__ bool keep(int e) const { return dr.mask ? dr.mask[(size_t)e * h + head] > dr.drop : true; }
};

// ======================================================================================================================
// a wave per row (64-edge tiles; sw / sc: the wave's 64-float / 64-int LDS scratch)
// ======================================================================================================================
template <class C>
__device__ __forceinline__ void gat_fwd_row_wave(const GatTrain &a, int r, int lane, float *sw, int *sc) {
  const int gid = lane / C::G, gl = lane % C::G;
  const int lb = a.row_ptr[r], deg = a.row_ptr[r + 1] - lb;
  const float ar = a.attn_row[a.nh(r)];
  Frag<C> acc;
  frag_zero<C>(acc);
  float m_run = -INFINITY, l_run = 0.f;
  for (int t0 = 0; t0 < deg; t0 += kWave) {
    const int nt = min(kWave, deg - t0);
    float s = -INFINITY;
    int c = 0;
    bool keep = true;
    if (lane < nt) {
      c = a.col_ind[lb + t0 + lane];
      s = leaky_relu(ar + a.attn_col[a.nh(c)], a.slope);
      keep = a.keep(lb + t0 + lane);
    }
    sc[lane] = c;
    online_step<C>(s, lane, sw, acc, m_run, l_run);  // the row sum counts every edge, dropped or not
    if (!keep) sw[lane] = 0.f;
    wave_sync();
    spmm_accum<C>(acc, sw, sc, nt, a.Xh, a.hf, a.f, gid, gl);
    wave_sync();
  }
  const float inv = (l_run != 0.f) ? a.dr.scale / l_run : 0.f;
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, inv, a.outh + (size_t)r * a.hf, a.f, gl);
  if (lane == 0) {
    a.edge_max[a.nh(r)] = deg > 0 ? m_run : -1e38f;  // the reference's sentinel (fused_gatconv_kernel.cu:46, 66)
    a.edge_sum[a.nh(r)] = l_run;
  }
}

template <class C>
__device__ __forceinline__ void gat_bwd_row_wave(const GatTrain &a, int r, int lane, float *sw, int *sc) {
  const int gid = lane / C::G, gl = lane % C::G;
  const int lb = a.row_ptr[r], deg = a.row_ptr[r + 1] - lb;
  float rs = 0.f;
  if (deg > 0) {
    const float ar = a.attn_row[a.nh(r)];
    const float mx = a.edge_max[a.nh(r)], inv = 1.f / a.edge_sum[a.nh(r)];
    Frag<C> go;
    frag_load<C>(go, a.dOh + (size_t)r * a.hf, a.f, gl);
    // sweep 1: g_e and t = sum_e P_e g_e.  Single-tile rows keep (P, g, slope factor) in registers; longer rows
    // park g_e in grad_edge and recompute P in sweep 2.
    float t = 0.f, p_keep = 0.f, g_keep = 0.f, lr_keep = 0.f;
    for (int t0 = 0; t0 < deg; t0 += kWave) {
      const int nt = min(kWave, deg - t0);
      int c = 0;
      if (lane < nt) c = a.col_ind[lb + t0 + lane];
      sc[lane] = c;
      wave_sync();
      tile_dots<C>(go, sc, nt, a.Xh, a.hf, a.f, gid, gl, sw);
      wave_sync();
      if (lane < nt) {
        const int e = lb + t0 + lane;
        const float pre = ar + a.attn_col[a.nh(c)];
        const float p = fast_exp(leaky_relu(pre, a.slope) - mx) * inv;
        const float ge = a.keep(e) ? sw[lane] * a.dr.scale : 0.f;
        t = fmaf(p, ge, t);
        p_keep = p;
        g_keep = ge;
        lr_keep = pre > 0.f ? 1.f : a.slope;
        if (deg > kWave) a.G_h[e] = ge;
      }
      wave_sync();
    }
    t = lanes_sum<kWave>(t);
    if (deg <= kWave) {
      float ge = 0.f;
      if (lane < deg) {
        ge = p_keep * (g_keep - t) * lr_keep;
        a.G_h[lb + lane] = ge;
      }
      rs = lanes_sum<kWave>(ge);
    } else {
      for (int e = lb + lane; e < lb + deg; e += kWave) {
        const float pre = ar + a.attn_col[a.nh(a.col_ind[e])];
        const float p = fast_exp(leaky_relu(pre, a.slope) - mx) * inv;
        const float ge = p * (a.G_h[e] - t) * (pre > 0.f ? 1.f : a.slope);  // this lane parked G_h[e] in sweep 1
        a.G_h[e] = ge;
        rs += ge;
      }
      rs = lanes_sum<kWave>(rs);
    }
  }
  if (lane == 0) a.grad_row[a.nh(r)] = rs;
}

template <class C>
__device__ __forceinline__ void gat_bwd_col_wave(const GatTrain &a, int j, int lane, float *sw, int *sc) {
  const int gid = lane / C::G, gl = lane % C::G;
  const int lb = a.col_ptr[j], n = a.col_ptr[j + 1] - lb;
  const float ac = a.attn_col[a.nh(j)];
  Frag<C> acc;
  frag_zero<C>(acc);
  float gs = 0.f;
  for (int t0 = 0; t0 < n; t0 += kWave) {
    const int nt = min(kWave, n - t0);
    float w = 0.f;
    int i = 0;
    if (lane < nt) {
      i = a.row_ind[lb + t0 + lane];
      const int e = a.permute[lb + t0 + lane];
      const float p = fast_exp(leaky_relu(a.attn_row[a.nh(i)] + ac, a.slope) - a.edge_max[a.nh(i)]) / a.edge_sum[a.nh(i)];
      w = a.keep(e) ? p * a.dr.scale : 0.f;
      gs += a.G_h[e];
    }
    sw[lane] = w;
    sc[lane] = i;
    wave_sync();
    spmm_accum<C>(acc, sw, sc, nt, a.dOh, a.hf, a.f, gid, gl);
    wave_sync();
  }
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, 1.f, a.gfeath + (size_t)j * a.hf, a.f, gl);
  gs = lanes_sum<kWave>(gs);
  if (lane == 0) a.grad_col[a.nh(j)] = gs;
}

// ======================================================================================================================
// a group of G lanes (one feature row wide) per row: low-degree graphs (molecules, peptides: ~2 edges per row), where a
// wave per row wastes most of its lanes.  As in gt_lowdeg.hip: EPW rows per wave, everything in registers, loops to
// each group's own degree.
// ======================================================================================================================
// Sum over the EPW lane groups of a wave (every group ends up with the result).
template <class C>
__device__ __forceinline__ float gat_groups_sum(float v) {
#pragma unroll
  for (int o = C::G; o < kWave; o <<= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// COOP: the row is taken by all EPW groups of the wave together (group gid: edges gid, gid + EPW, ...), partial
// results merged across the groups -- for the few long rows of an otherwise low-degree graph.
template <class C, bool COOP>
__device__ __forceinline__ void gat_fwd_row_group(const GatTrain &a, int r, int gid, int gl) {
  const int lb = a.row_ptr[r], deg = a.row_ptr[r + 1] - lb;
  const float ar = a.attn_row[a.nh(r)];
  Frag<C> acc;
  frag_zero<C>(acc);
  float m_run = -INFINITY, l_run = 0.f;  // online softmax: one sweep, one dependent gather chain per edge
  for (int e = COOP ? gid : 0; e < deg; e += COOP ? C::EPW : 1) {
    const int c = a.col_ind[lb + e];
    Frag<C> x;
    frag_load<C>(x, a.Xh + (size_t)c * a.hf, a.f, gl);
    const float s = leaky_relu(ar + a.attn_col[a.nh(c)], a.slope);
    const float m_new = fmaxf(m_run, s);
    const float sc = (m_run == -INFINITY) ? 0.f : fast_exp(m_run - m_new);
    const float p = fast_exp(s - m_new);
    l_run = l_run * sc + p;  // the row sum counts every edge, dropped or not
    frag_scale<C>(acc, sc);
    frag_fma<C>(acc, a.keep(lb + e) ? p : 0.f, x);
    m_run = m_new;
  }
  if constexpr (COOP) {  // merge the groups' (max, sum, accumulator) states pairwise
#pragma unroll
    for (int o = C::G; o < kWave; o <<= 1) {
      const float m_o = __shfl_xor(m_run, o, kWave), l_o = __shfl_xor(l_run, o, kWave);
      const float m_new = fmaxf(m_run, m_o);
      const float sa = (m_run == -INFINITY) ? 0.f : fast_exp(m_run - m_new);
      const float sb = (m_o == -INFINITY) ? 0.f : fast_exp(m_o - m_new);
      l_run = l_run * sa + l_o * sb;
#pragma unroll
      for (int ch = 0; ch < C::NCH; ++ch)
#pragma unroll
        for (int k = 0; k < C::VEC; ++k)
          acc.v[ch][k] = acc.v[ch][k] * sa + __shfl_xor(acc.v[ch][k], o, kWave) * sb;
      m_run = m_new;
    }
  }
  if (!COOP || gid == 0) {
    frag_store_scaled<C>(acc, l_run != 0.f ? a.dr.scale / l_run : 0.f, a.outh + (size_t)r * a.hf, a.f, gl);
    if (gl == 0) {
      a.edge_max[a.nh(r)] = deg > 0 ? m_run : -1e38f;
      a.edge_sum[a.nh(r)] = l_run;
    }
  }
}

template <class C, bool COOP>
__device__ __forceinline__ void gat_bwd_row_group(const GatTrain &a, int r, int gid, int gl) {
  const int lb = a.row_ptr[r], deg = a.row_ptr[r + 1] - lb;
  const int e0 = COOP ? gid : 0, es = COOP ? C::EPW : 1;
  float rs = 0.f;
  if (deg > 0) {
    const float ar = a.attn_row[a.nh(r)];
    const float mx = a.edge_max[a.nh(r)], inv = 1.f / a.edge_sum[a.nh(r)];
    Frag<C> go;
    frag_load<C>(go, a.dOh + (size_t)r * a.hf, a.f, gl);
    float t = 0.f;
    for (int e = e0; e < deg; e += es) {  // g_e parked in grad_edge by lane 0 (every lane of the group holds the value)
      const int c = a.col_ind[lb + e];
      Frag<C> x;
      frag_load<C>(x, a.Xh + (size_t)c * a.hf, a.f, gl);
      const float dp = lanes_sum<C::G>(frag_dot<C>(go, x));
      const float ge = a.keep(lb + e) ? dp * a.dr.scale : 0.f;
      t = fmaf(fast_exp(leaky_relu(ar + a.attn_col[a.nh(c)], a.slope) - mx) * inv, ge, t);
      if (gl == 0) a.G_h[lb + e] = ge;
    }
    if constexpr (COOP) t = gat_groups_sum<C>(t);
    if (gl == 0)  // lane 0 of the group re-reads what it parked
      for (int e = e0; e < deg; e += es) {
        const float pre = ar + a.attn_col[a.nh(a.col_ind[lb + e])];
        const float ge = fast_exp(leaky_relu(pre, a.slope) - mx) * inv * (a.G_h[lb + e] - t) * (pre > 0.f ? 1.f : a.slope);
        a.G_h[lb + e] = ge;
        rs += ge;
      }
  }
  if constexpr (COOP) rs = gat_groups_sum<C>(rs);  // (lanes gl != 0 hold 0; lane 0 of group 0 gets the total)
  if (gl == 0 && (!COOP || gid == 0)) a.grad_row[a.nh(r)] = rs;
}

template <class C, bool COOP>
__device__ __forceinline__ void gat_bwd_col_group(const GatTrain &a, int j, int gid, int gl) {
  const int lb = a.col_ptr[j], n = a.col_ptr[j + 1] - lb;
  const float ac = a.attn_col[a.nh(j)];
  Frag<C> acc;
  frag_zero<C>(acc);
  float gs = 0.f;
  for (int t = COOP ? gid : 0; t < n; t += COOP ? C::EPW : 1) {
    const int i = a.row_ind[lb + t], e = a.permute[lb + t];
    const float p = fast_exp(leaky_relu(a.attn_row[a.nh(i)] + ac, a.slope) - a.edge_max[a.nh(i)]) / a.edge_sum[a.nh(i)];
    Frag<C> go;
    frag_load<C>(go, a.dOh + (size_t)i * a.hf, a.f, gl);
    frag_fma<C>(acc, a.keep(e) ? p * a.dr.scale : 0.f, go);
    gs += a.G_h[e];
  }
  if constexpr (COOP) {
    frag_reduce_groups<C>(acc);
    gs = gat_groups_sum<C>(gs);
  }
  if (!COOP || gid == 0) {
    frag_store_scaled<C>(acc, 1.f, a.gfeath + (size_t)j * a.hf, a.f, gl);
    if (gl == 0) a.grad_col[a.nh(j)] = gs;
  }
}

// ======================================================================================================================
// kernels.  PASS: 0 forward, 1 backward CSR pass, 2 backward CSC pass.
// ======================================================================================================================
template <class C, int PASS>
__device__ __forceinline__ void gat_wave_pass(const GatTrain &a, int r, int lane, float *sw, int *sc) {
  if constexpr (PASS == 0) gat_fwd_row_wave<C>(a, r, lane, sw, sc);
  else if constexpr (PASS == 1) gat_bwd_row_wave<C>(a, r, lane, sw, sc);
  else gat_bwd_col_wave<C>(a, r, lane, sw, sc);
}
template <class C, int PASS, bool COOP>
__device__ __forceinline__ void gat_group_pass(const GatTrain &a, int r, int gid, int gl) {
  if constexpr (PASS == 0) gat_fwd_row_group<C, COOP>(a, r, gid, gl);
  else if constexpr (PASS == 1) gat_bwd_row_group<C, COOP>(a, r, gid, gl);
  else gat_bwd_col_group<C, COOP>(a, r, gid, gl);
}

// general: a wave per row / column over the whole graph (grid-strided) or over the ranges of `rl`
template <class C, int PASS>
__global__ __launch_bounds__(kBlock) void gat_train_wave_kernel(GatTrain a, RowLists rl) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  a.head = blockIdx.y;
  a.Xh += (size_t)a.head * a.f;
  a.dOh += (size_t)a.head * a.f;
  if (a.outh) a.outh += (size_t)a.head * a.f;
  if (a.gfeath) a.gfeath += (size_t)a.head * a.f;
  if (a.G_h) a.G_h += (size_t)a.head * a.nnz;
  int beg, end, step;
  row_span(rl, a.m, wave, beg, end, step);  // (CSC pass: closed ranges, a range's columns are its rows)
  for (int r = beg; r < end; r += step) gat_wave_pass<C, PASS>(a, r, lane, sw, sc);
}

// low-degree graphs: a workgroup takes blocks of kBlock / G consecutive rows, one lane group per row -- unless a wave's
// EPW rows include one of more than kGatGroupMaxDegree entries, which a single lane group would walk serially while the
// rest of the wave waits (a hub of a citation graph): that wave takes its rows one after the other with all its groups
// on each (COOP).  The choice is wave-uniform (ballot): no barrier, no LDS.
constexpr int kGatGroupMaxDegree = 24;
template <class C, int PASS>
__global__ __launch_bounds__(kBlock) void gat_train_group_kernel(GatTrain a) {
  constexpr int G = C::G, R = kBlock / G;  // rows per block
  const int gid = (threadIdx.x & (kWave - 1)) / G, gl = threadIdx.x % G, wave = threadIdx.x / kWave;
  a.head = blockIdx.y;
  a.Xh += (size_t)a.head * a.f;
  a.dOh += (size_t)a.head * a.f;
  if (a.outh) a.outh += (size_t)a.head * a.f;
  if (a.gfeath) a.gfeath += (size_t)a.head * a.f;
  if (a.G_h) a.G_h += (size_t)a.head * a.nnz;
  const int *ptr = PASS == 2 ? a.col_ptr : a.row_ptr;
  for (int b0 = blockIdx.x * R; b0 < a.m; b0 += gridDim.x * R) {
    const int r = b0 + threadIdx.x / G;
    const int deg = r < a.m ? ptr[r + 1] - ptr[r] : 0;
    if (__any(deg > kGatGroupMaxDegree)) {
      for (int rr = b0 + wave * C::EPW; rr < min(a.m, b0 + (wave + 1) * C::EPW); ++rr)
        gat_group_pass<C, PASS, true>(a, rr, gid, gl);
    } else if (r < a.m) {
      gat_group_pass<C, PASS, false>(a, r, gid, gl);
    }
  }
}

// DFGNN_LOWDEG=0 in the environment (diagnostic switch, read once) keeps low-degree graphs on the wave-per-row kernels
static bool lowdeg_enabled() {
  static const bool on = [] { const char *e = getenv("DFGNN_LOWDEG"); return !e || atoi(e) != 0; }();
  return on;
}

static dim3 group_grid(int m, int h, int G) {
  const long per = kBlock / G;
  long blocks = ((long)m + per - 1) / per;
  if (blocks > 16384) blocks = 16384;
  return dim3((unsigned)(blocks < 1 ? 1 : blocks), h);
}
static dim3 wave_grid(int m, int h, const RowLists &rl) {
  if (rl.na + rl.nb > 0) return dim3(rl.na + rl.nb, h);
  const long want = ((long)m + kWavesPerBlock - 1) / kWavesPerBlock;
  return dim3((unsigned)(want > (1 << 20) ? (1 << 20) : want), h);
}
// the ranges of a plan the matrix-core kernels do not serve (none: the whole graph)
static inline RowLists rest_of(const Plan *p) {
  if (!p) return RowLists{nullptr, nullptr, 0, 0};
  return RowLists{p->fit() + 2 * (size_t)p->num_dense, p->spill(), p->num_fit - p->num_dense, p->num_spill};
}

template <int PASS>
static int launch_gat_train_pass(const GatTrain &a, bool v4, const Plan *rest, hipStream_t s) {
  const RowLists rl = rest_of(rest);
  const bool groups = !rest && lowdeg_enabled() && low_degree(a.m, a.nnz);
  return dispatch_cfg(a.f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    if (groups) gat_train_group_kernel<C, PASS><<<group_grid(a.m, a.h, C::G), kBlock, 0, s>>>(a);
    else gat_train_wave_kernel<C, PASS><<<wave_grid(a.m, a.h, rl), kBlock, 0, s>>>(a, rl);
    return launch_status();
  });
}

static GatTrain gat_args(const Csr &g, const float *attn_row, const float *attn_col, float slope,
                         const float *edge_mask, float attn_drop) {
  GatTrain a{};
  a.m = g.m; a.nnz = g.nnz; a.h = g.h; a.f = g.f; a.hf = (size_t)g.h * g.f;
  a.row_ptr = g.row_ptr; a.col_ind = g.col_ind;
  a.attn_row = attn_row; a.attn_col = attn_col; a.slope = slope;
  a.dr = GatDrop{edge_mask, attn_drop, edge_mask ? 1.f / (1.f - attn_drop) : 1.f};
  return a;
}

int launch_gat_train_fwd(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                         const float *edge_mask, float attn_drop, float *edge_max, float *edge_sum, float *out,
                         hipStream_t s, const Plan *rest) {
  GatTrain a = gat_args(g, attn_row, attn_col, slope, edge_mask, attn_drop);
  a.Xh = X; a.dOh = X; a.outh = out; a.edge_max = edge_max; a.edge_sum = edge_sum;
  return launch_gat_train_pass<0>(a, (g.f % 4 == 0) && aligned16(X) && aligned16(out), rest, s);
}

int launch_gat_bwd_rows(const Csr &g, const float *attn_row, const float *attn_col, float slope, const float *X,
                        const float *edge_max, const float *edge_sum, const float *edge_mask, float attn_drop,
                        const float *grad_out, float *grad_edge, float *grad_row, hipStream_t s, const Plan *rest) {
  GatTrain a = gat_args(g, attn_row, attn_col, slope, edge_mask, attn_drop);
  a.Xh = X; a.dOh = grad_out; a.G_h = grad_edge; a.grad_row = grad_row;
  a.edge_max = const_cast<float *>(edge_max); a.edge_sum = const_cast<float *>(edge_sum);
  return launch_gat_train_pass<1>(a, (g.f % 4 == 0) && aligned16(X) && aligned16(grad_out), rest, s);
}

int launch_gat_bwd_cols(const Csr &g, const int *col_ptr, const int *row_ind, const int *permute,
                        const float *attn_row, const float *attn_col, float slope, const float *edge_max,
                        const float *edge_sum, const float *edge_mask, float attn_drop, const float *grad_edge,
                        const float *grad_out, float *grad_feat, float *grad_col, hipStream_t s, const Plan *rest) {
  GatTrain a = gat_args(g, attn_row, attn_col, slope, edge_mask, attn_drop);
  a.col_ptr = col_ptr; a.row_ind = row_ind; a.permute = permute;
  a.Xh = grad_out; a.dOh = grad_out; a.gfeath = grad_feat; a.G_h = const_cast<float *>(grad_edge); a.grad_col = grad_col;
  a.edge_max = const_cast<float *>(edge_max); a.edge_sum = const_cast<float *>(edge_sum);
  return launch_gat_train_pass<2>(a, (g.f % 4 == 0) && aligned16(grad_out) && aligned16(grad_feat), rest, s);
}

}  // namespace dfgnn
