// gat_tiling_chunked.hip -- GAT 'tiling' for super-node full graphs whose feature table does not fit the L2s: column
// chunks pinned per XCD, per-(row, chunk) online-softmax partial states, a merge pass.
//
// The reference's tiling kernel (fused_gatconv_tiling.cu:9-76) and gat_tiling_fwd_kernel (gat_fwd.hip) give a row to one
// block / wave, which walks the row's neighbours in tiles and gathers X[col] (f floats) for each: on the reddit-like graph
// (232 965 nodes, 114.6 M edges, f = 128) that is 58.7 GB of 512-byte row gathers from a 119 MB table for 0.70 GB of
// compulsory traffic.  The table lives in the Infinity Cache, not in the 8 x 4 MiB L2s, and random rows come out of it at
// ~7 TB/s (MI355X_MICROARCH.md, "Indexed rows"): 8.3 ms, the ceiling of that ALGORITHM, whatever the kernel does.
// Rows served from an XCD's own L2 arrive 2.3 x faster (17 - 19 TB/s).  So the columns are cut into chunks of
// `chunk_rows` nodes whose feature rows fit one L2 (8192 x 512 B = 4 MiB), and the work is re-ordered chunk-major:
//   * preprocessing (once per graph, cached by the binding like a block plan): the edges sorted by (chunk of the column,
//     row) -- seg_ptr[c m + r] .. seg_ptr[c m + r + 1] are the edges of row r into chunk c -- with the column kept as a
//     16-bit offset inside its chunk (half the index bytes);
//   * gat_chunk_partial_kernel: a work item = (chunk, 64 consecutive rows).  Items are queued PER XCD (chunk c belongs
//     to XCD c mod 8, its items in chunk order) and a workgroup pulls from the queue of the XCD it actually runs on
//     (HW_REG_XCC_ID), so at any time an XCD's workgroups gather from one chunk = 4 MiB = its L2; an empty queue steals
//     from the next one (the tail).  A wave runs the online softmax over a (row, chunk) segment and stores the partial
//     state (max, sum, un-normalised accumulator) -- the flash-attention split-K form;
//   * gat_chunk_merge_kernel: a wave per row merges its <= nchunks partial states: out = sum_c e^(m_c - M) acc_c /
//     sum_c e^(m_c - M) l_c.
// Extra traffic: (2 + f) floats per non-empty (row, chunk) pair, written once and read once (reddit-like: 2 x 3.3 GB at the
// HBM rate, ~1.2 ms) against 58.7 GB of gathers moving from the Infinity-Cache rate to the L2 rate.  Super-node rows
// (degree 20 k) are split over the chunks by construction.  Placement is for speed only: nothing depends on which XCD
// runs what (the queues are plain atomic counters, re-zeroed by a memset node ahead of every launch).
#include "../../include/dfgnn.h"
#include "dfgnn_launch.hpp"
#include "dfgnn_rows.hpp"

namespace dfgnn {

constexpr int kChunkItemRows = 64;  // rows of a work item (16 per wave)
constexpr int kXcds = 8;

struct ChunkGraph {
  int m, h, f, chunk_rows, nchunks;
  const int *seg_ptr;          // [nchunks * m + 1]
  const short *ccol;           // [nnz] column - chunk * chunk_rows, chunk-major edge order
};

// One (row, chunk) segment by one wave: online softmax over 64-edge tiles (gat_row_online's loop) without the final
// normalisation.  sw / sc: this wave's 64-float / 64-int LDS scratch.
template <class C>
__device__ __forceinline__ void gat_segment_online(int eb, int ee, const short *__restrict__ ccol, int col0, float ar,
                                                   const float *__restrict__ attn_col_h, int h, float slope,
                                                   const float *__restrict__ Xh, size_t hf, int f, float *sw, int *sc,
                                                   float *__restrict__ pmax, float *__restrict__ psum,
                                                   float *__restrict__ pacc, int lane) {
  const int gid = lane / C::G, gl = lane % C::G;
  Frag<C> acc;
  frag_zero<C>(acc);
  float m_run = -INFINITY, l_run = 0.f;
  for (int t0 = eb; t0 < ee; t0 += kWave) {
    const int nt = min(kWave, ee - t0);
    float s = -INFINITY;
    int c = col0;
    if (lane < nt) {
      c = col0 + (int)ccol[t0 + lane];
      s = leaky_relu(ar + attn_col_h[(size_t)c * h], slope);
    }
    sc[lane] = c;
    online_step<C>(s, lane, sw, acc, m_run, l_run);
    wave_sync();
    spmm_accum<C>(acc, sw, sc, nt, Xh, hf, f, gid, gl);
    wave_sync();
  }
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, 1.f, pacc, f, gl);
  if (lane == 0) {
    *pmax = m_run;
    *psum = l_run;
  }
}

// The same with the FIRST tile's column offsets (c0: this lane's edge of the tile, -1 = none) and attn_col values (a0)
// already in registers: the kernel below fetches them one / two segments ahead (seg_ptr -> ccol -> attn_col -> X is a chain
// of four dependent memory round trips per segment, and a segment is ~30 edges: without the look-ahead the chain, not the
// gather rate, is what a wave's time is made of).
template <class C>
__device__ __forceinline__ void gat_segment_online_pre(int eb, int ee, int c0, float a0, const short *__restrict__ ccol, int col0,
                                                       float ar, const float *__restrict__ attn_col_h, int h, float slope,
                                                       const float *__restrict__ Xh, size_t hf, int f, float *sw, int *sc,
                                                       float *__restrict__ pmax, float *__restrict__ psum,
                                                       float *__restrict__ pacc, int lane) {
  const int gid = lane / C::G, gl = lane % C::G;
  Frag<C> acc;
  frag_zero<C>(acc);
  float m_run = -INFINITY, l_run = 0.f;
  for (int t0 = eb; t0 < ee; t0 += kWave) {
    const int nt = min(kWave, ee - t0);
    float s = -INFINITY;
    int c = col0;
    if (t0 == eb) {  // (wave-uniform)
      if (lane < nt) {
        c = col0 + c0;
        s = leaky_relu(ar + a0, slope);
      }
    } else if (lane < nt) {
      c = col0 + (int)ccol[t0 + lane];
      s = leaky_relu(ar + attn_col_h[(size_t)c * h], slope);
    }
    sc[lane] = c;
    online_step<C>(s, lane, sw, acc, m_run, l_run);
    wave_sync();
    spmm_accum<C>(acc, sw, sc, nt, Xh, hf, f, gid, gl);
    wave_sync();
  }
  frag_reduce_groups<C>(acc);
  if (gid == 0) frag_store_scaled<C>(acc, 1.f, pacc, f, gl);
  if (lane == 0) {
    *pmax = m_run;
    *psum = l_run;
  }
}

template <class C>
__global__ __launch_bounds__(kBlock) void gat_chunk_partial_kernel(ChunkGraph g, const float *__restrict__ attn_row,
                                                                   const float *__restrict__ attn_col, float slope,
                                                                   const float *__restrict__ X, float *__restrict__ pmax,
                                                                   float *__restrict__ psum, float *__restrict__ pacc,
                                                                   int *__restrict__ heads) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * kScratchFloatsPerWave];
  __shared__ int s_item;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  float *sw = lds + wave * kScratchFloatsPerWave;
  int *sc = reinterpret_cast<int *>(sw + kWave);
  const int nblk = (g.m + kChunkItemRows - 1) / kChunkItemRows;
  const size_t hf = (size_t)g.h * g.f;
  const int xcc = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & (kXcds - 1));  // HW_REG_XCC_ID: the XCD this runs on
  for (int q = 0; q < kXcds; ++q) {  // own queue first, then the others' leftovers
    const int x = (xcc + q) & (kXcds - 1);
    const int nch_x = (g.nchunks > x) ? (g.nchunks - x + kXcds - 1) / kXcds : 0;  // chunks x, x + 8, ...
    const int total = nch_x * g.h * nblk;
    for (;;) {
      if (threadIdx.x == 0) s_item = (total > 0) ? atomicAdd(&heads[x * 8], 1) : total;  // (a 32-byte slot per queue)
      __syncthreads();
      const int item = s_item;
      __syncthreads();
      if (item >= total) break;
      // items of a queue: chunk-major (all heads and row blocks of chunk x, then of chunk x + 8, ...)
      const int k = item / nblk, rb = item - k * nblk;
      const int chunk = x + kXcds * (k / g.h), head = k % g.h;
      const float *Xh = X + (size_t)head * g.f, *acol_h = attn_col + head;
      const int *sp = g.seg_ptr + (size_t)chunk * g.m;
      const int r_end = min(g.m, (rb + 1) * kChunkItemRows);
      // this wave's rows of the item: rbase + kWavesPerBlock k, k < NR.  Their segment bounds and attn_row values come with
      // ONE load each (lane l < NR: seg_ptr[r_l], lane NR + l: seg_ptr[r_l + 1]; rows past the end read as empty), the
      // first tile's column offsets are fetched two segments ahead and its attn_col values one segment ahead.
      constexpr int NR = kChunkItemRows / kWavesPerBlock;
      static_assert(2 * NR <= kWave, "segment bounds of a wave's rows in one load");
      const int rbase = rb * kChunkItemRows + wave;
      int spv = 0;
      float arv = 0.f;
      {
        const int rr = rbase + kWavesPerBlock * (lane & (NR - 1));
        const bool ok = rr < r_end && lane < 2 * NR;
        spv = ok ? sp[rr + (lane >= NR ? 1 : 0)] : 0;
        if (lane < NR) arv = (rr < r_end) ? attn_row[(size_t)rr * g.h + head] : 0.f;
      }
      const int col0 = chunk * g.chunk_rows;
      auto first_cols = [&](int k) -> int {  // this lane's column offset in the first tile of segment k, -1 = none
        if (k >= NR) return -1;
        const int eb = __builtin_amdgcn_readlane(spv, k), ee = __builtin_amdgcn_readlane(spv, NR + k);
        return (lane < min(kWave, ee - eb)) ? (int)g.ccol[eb + lane] : -1;
      };
      auto first_attn = [&](int c) -> float { return (c >= 0) ? acol_h[(size_t)(col0 + c) * g.h] : 0.f; };
      int c0 = first_cols(0), c1 = first_cols(1);
      float a0 = first_attn(c0);
      for (int k = 0; k < NR; ++k) {
        const int c2 = first_cols(k + 2);
        const float a1 = first_attn(c1);
        const int eb = __builtin_amdgcn_readlane(spv, k), ee = __builtin_amdgcn_readlane(spv, NR + k);
        if (ee > eb) {  // (empty segments keep the zero the launcher's memset put into psum)
          const int r = rbase + kWavesPerBlock * k;
          const size_t slot = ((size_t)head * g.nchunks + chunk) * g.m + r;
          const float ar = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, arv), k));
          gat_segment_online_pre<C>(eb, ee, c0, a0, g.ccol, col0, ar, acol_h, g.h, slope, Xh, hf, g.f, sw, sc, pmax + slot,
                                    psum + slot, pacc + slot * g.f, lane);
        }
        c0 = c1;
        a0 = a1;
        c1 = c2;
      }
    }
  }
}

// out[r, head, :] = sum_c e^(m_c - M) acc_c / sum_c e^(m_c - M) l_c over the row's non-empty chunks (M = max_c m_c)
template <class C>
__global__ __launch_bounds__(kBlock) void gat_chunk_merge_kernel(ChunkGraph g, const float *__restrict__ pmax,
                                                                 const float *__restrict__ psum,
                                                                 const float *__restrict__ pacc, float *__restrict__ out) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int gid = lane / C::G, gl = lane % C::G;
  const int head = blockIdx.y;
  const size_t hf = (size_t)g.h * g.f;
  for (int r = blockIdx.x * kWavesPerBlock + wave; r < g.m; r += gridDim.x * kWavesPerBlock) {
    const size_t base = (size_t)head * g.nchunks * g.m + r;
    float M = -INFINITY;
    for (int c = lane; c < g.nchunks; c += kWave)
      if (psum[base + (size_t)c * g.m] != 0.f) M = fmaxf(M, pmax[base + (size_t)c * g.m]);
    M = lanes_max<kWave>(M);
    Frag<C> acc;
    frag_zero<C>(acc);
    float l = 0.f;
    for (int c = gid; c < g.nchunks; c += C::EPW) {
      const size_t slot = base + (size_t)c * g.m;
      const float ls = psum[slot];
      if (ls != 0.f) {
        const float w = fast_exp(pmax[slot] - M);
        Frag<C> a;
        frag_load<C>(a, pacc + slot * g.f, g.f, gl);
        frag_fma<C>(acc, w, a);
        l = fmaf(w, ls, l);
      }
    }
    // every lane of a group holds the same l: sum over the groups (one lane per group), then over the accumulators
    float lt = (gl == 0) ? l : 0.f;
    lt = lanes_sum<kWave>(lt);
    frag_reduce_groups<C>(acc);
    if (gid == 0) frag_store_scaled<C>(acc, (lt != 0.f) ? 1.f / lt : 0.f, out + (size_t)r * hf + (size_t)head * g.f, g.f, gl);
  }
}

}  // namespace dfgnn

using namespace dfgnn;

extern "C" {

// workspace: 8 queue heads (256 B), then psum, pmax [h nchunks m], then (16-byte aligned) pacc [h nchunks m f]
static size_t chunked_pacc_off(size_t slots) { return (256 + slots * 8 + 15) & ~(size_t)15; }
size_t dfgnn_gat_tiling_chunked_ws_bytes(int m, int h, int f, int chunk_rows) {
  if (m <= 0 || h <= 0 || f <= 0 || chunk_rows <= 0) return 0;
  const size_t nchunks = ((size_t)m + chunk_rows - 1) / chunk_rows, slots = (size_t)h * nchunks * m;
  return chunked_pacc_off(slots) + slots * 4 * (size_t)f;
}

int dfgnn_gat_tiling_chunked_fwd(int m, int nnz, int h, int f, int chunk_rows, const int *seg_ptr, const short *ccol,
                                 const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                                 float *out, void *ws, size_t ws_bytes, dfgnn_stream_t stream) {
  if (m < 0 || nnz < 0 || h < 0 || f < 0 || chunk_rows <= 0 || chunk_rows > 32768) return kErrBadArg;
  if (m == 0 || h == 0 || f == 0) return 0;
  if (h > 65535) return kErrUnsupported;
  if (!seg_ptr || (nnz > 0 && !ccol) || !attn_row || !attn_col || !X || !out || !ws) return kErrBadArg;
  if (ws_bytes < dfgnn_gat_tiling_chunked_ws_bytes(m, h, f, chunk_rows)) return kErrBadArg;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nchunks = (m + chunk_rows - 1) / chunk_rows;
  const size_t slots = (size_t)h * nchunks * m;
  if (slots * (size_t)f >= ((size_t)1 << 40)) return kErrUnsupported;
  int *heads = reinterpret_cast<int *>(ws);
  float *psum = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + 256), *pmax = psum + slots;
  float *pacc = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + chunked_pacc_off(slots));
  // queue heads and the "empty segment" marks (psum = 0): one memset node ahead of the launch (graph-capturable)
  if (hipError_t rc = hipMemsetAsync(ws, 0, 256 + slots * 4, s)) return (int)rc;
  const ChunkGraph g{m, h, f, chunk_rows, nchunks, seg_ptr, ccol};
  const bool v4 = (f % 4 == 0) && aligned16(X) && aligned16(out) && aligned16(pacc);
  return dispatch_cfg(f, v4, [&](auto cfg) {
    using C = decltype(cfg);
    // enough workgroups to fill every CU 8 times over; each pulls items until the queues are dry
    gat_chunk_partial_kernel<C><<<dim3(256 * 8), kBlock, 0, s>>>(g, attn_row, attn_col, negative_slope, X, pmax, psum, pacc, heads);
    if (int rc = launch_status()) return rc;
    const dim3 grid((unsigned)min((long)65535, ((long)m + kWavesPerBlock - 1) / kWavesPerBlock), h);
    gat_chunk_merge_kernel<C><<<grid, kBlock, 0, s>>>(g, pmax, psum, pacc, out);
    return launch_status();
  });
}

}  // extern "C"
