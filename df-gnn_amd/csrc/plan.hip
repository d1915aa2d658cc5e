// plan.hip -- block plan of a CSR graph: which contiguous node ranges are "closed" (every edge of
// a row in the range lands on a column in the range) and small enough for one workgroup to keep the
// range's K / V (GAT: X) rows resident in LDS.
//
// A DGL batch of small graphs is block-diagonal (reference: GraphDataLoader batches in
// DFGNN/script/test/test_batch_graph.py:67-71), so its closed ranges are the member
// graphs.  The reference leaves the K/V gathers of such batches to the L2 (fused_gtconv_hyper.cu:
// 333-337, 399-409); on MI355X one graph's K and V rows fit the 160 KB LDS of a CU, so the hyper
// kernels run one workgroup per closed range with the gathers served from LDS.  Ranges that do not
// fit (a full graph such as cora/reddit is a single closed range) are cut into <= 16-row "spill"
// chunks that run the general row-block kernel.
//
// Plan buffer (int32, device), sized by dfgnn_plan_ints(m):
//   [0 .. 12)           header: num_fit, num_spill, max_fit_nodes, max_fit_edges, m, nnz, f, budget,
//                               num_edge_global (fit ranges whose per-edge fp32 array lives in global scratch),
//                               num_dense (fit ranges that qualify for the matrix-core kernels; they come first), 0, 0
//   [12 .. 12+2m)       fit ranges   (n0, n1 | kPlanEdgeGlobal | kPlanDense) pairs: dense ones first, each group
//                                    largest first
//   [12+2m .. 12+4m)    spill chunks (r0, r1) pairs, r1 - r0 <= kHyperRows
//   [12+4m .. )         scratch: pairs[m] (cover, bad) as int2, bounds[m+1], count, rocPRIM temporary storage
//   [.. behind coords)  mask[8 m], maskT[8 m]: edge bitmaps of the dense ranges, 8 words per node (plan_coords_kernel;
//                       offset: dfgnn_launch.hpp:plan_mask_off) -- for the kernels that need the edge SET only
//   [hdr[11] .. )       coords: uint16[nnz], for every edge e of a dense range (i - n0) << 8 | (j - n0) -- its row and
//                       column within the range (both < 256).  The matrix-core kernels read these 2 bytes per edge
//                       instead of rows[e] and col_ind[e] (8 bytes): the sparse structure of a dense range costs a
//                       quarter of the bytes, and the prologue that waits for it a quarter of the transfer.
//
// Build (all on the GPU, four steps, no host round trip before the final header copy):
//   1. plan_row_extent_kernel  a 16-lane group per row: column extent [lo, hi] of the row, duplicate-edge check;
//                              the row blocks every cut lo <= i < hi between rows i and i+1: diff[lo] += 1,
//                              diff[hi] -= 1 (integer atomics)
//   2. rocprim::inclusive_scan of (diff, has-duplicate) pairs  ->  cover[i] = rows blocking cut i, bad[i] = rows with a
//                              duplicate edge up to i
//   3. rocprim::select         the open cuts (cover == 0) = ends of the natural closed ranges, in order
//   4. plan_cut_kernel         one workgroup: greedy merge of the natural ranges (a serial walk over LDS copies),
//                              classification, largest-first sort
//   5. plan_coords_kernel      a workgroup per dense range (grid-stride), 16 lanes per row: the packed coordinates
//
// "Dense" (gt_dense.hip, dfgnn_dense.hpp): at most 255 nodes, at least one edge per 32 node pairs, f in {32, 64, 128}
// and no duplicate edge in any row of the range -- the one thing a dense mask cannot represent.
#include <cstring>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "../../include/dfgnn.h"
#include "dfgnn_launch.hpp"

namespace dfgnn {

constexpr int kPlanThreads = 1024;

struct PlanPair {
  int cover, bad;  // before the scan: (diff, row has a duplicate edge); after: inclusive prefix sums
};
struct PlanPairAdd {
  __host__ __device__ PlanPair operator()(const PlanPair &a, const PlanPair &b) const {
    return PlanPair{a.cover + b.cover, a.bad + b.bad};
  }
};
struct PlanCutOpen {  // flag iterator of the select: cut after row i is open
  const PlanPair *pairs;
  __host__ __device__ bool operator()(int i) const { return pairs[i].cover == 0; }
};

constexpr int kExtentLanes = 16;  // lanes per row in plan_row_extent_kernel

// pairs must be zero on entry.
__global__ __launch_bounds__(256) void plan_row_extent_kernel(int m, const int *__restrict__ row_ptr,
                                                              const int *__restrict__ col_ind, PlanPair *pairs) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) / kExtentLanes;
  const int gl = threadIdx.x % kExtentLanes;
  if (i >= m) return;  // (whole groups leave together: 256 % 16 == 0)
  const int ea = row_ptr[i], eb = row_ptr[i + 1];
  int cl = i, ch = i;
  for (int e = ea + gl; e < eb; e += kExtentLanes) {
    const int c = col_ind[e];
    cl = min(cl, c);
    ch = max(ch, c);
  }
#pragma unroll
  for (int o = 1; o < kExtentLanes; o <<= 1) {
    cl = min(cl, __shfl_xor(cl, o, kExtentLanes));
    ch = max(ch, __shfl_xor(ch, o, kExtentLanes));
  }
  // Duplicate columns in the row?  Only rows that could sit in a dense range matter (the range holds cl .. ch, so
  // its span and its length are below 256): every lane marks its columns in a 256-bit map; the row has a duplicate
  // iff the union of the maps has fewer bits than the row has edges.
  int bad = 1;
  if (ch - cl < 256 && eb - ea < 256) {
    unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    for (int e = ea + gl; e < eb; e += kExtentLanes) {
      const int k = col_ind[e] - cl;
      const unsigned long long bit = 1ull << (k & 63);
      const int q = k >> 6;
      w0 |= q == 0 ? bit : 0;
      w1 |= q == 1 ? bit : 0;
      w2 |= q == 2 ? bit : 0;
      w3 |= q == 3 ? bit : 0;
    }
#pragma unroll
    for (int o = 1; o < kExtentLanes; o <<= 1) {
      w0 |= __shfl_xor(w0, o, kExtentLanes);
      w1 |= __shfl_xor(w1, o, kExtentLanes);
      w2 |= __shfl_xor(w2, o, kExtentLanes);
      w3 |= __shfl_xor(w3, o, kExtentLanes);
    }
    bad = (__popcll(w0) + __popcll(w1) + __popcll(w2) + __popcll(w3) != eb - ea) ? 1 : 0;
  }
  if (gl == 0) {
    pairs[i].bad = bad;  // 1: the row has a duplicate edge (or is too wide / long to be part of a dense range)
    if (ch > cl) {       // the row blocks the cuts cl .. ch - 1
      atomicAdd(&pairs[cl].cover, 1);
      atomicAdd(&pairs[ch].cover, -1);
    }
  }
}

// LDS bytes of a range of n nodes holding ed edges: resident rows + 1/sum + rebased row_ptr + narrowed column ids,
// plus (full only) the per-edge fp32 array.  Layout: dfgnn_block.hpp:carve_block_lds.  (32-bit arithmetic on clamped
// sizes: anything clamped is far over the budget anyway.)
__device__ __forceinline__ int plan_lite_bytes(int n, int ed, int f) {
  n = min(n, 1 << 16);
  ed = min(ed, 1 << 24);
  return n * (4 * f + 8) + ed * (n <= 256 ? 1 : 2);
}
__device__ __forceinline__ int plan_full_bytes(int n, int ed, int f) { return plan_lite_bytes(n, ed, f) + 4 * min(ed, 1 << 24); }

constexpr int kPlanCache = 4096;  // natural ranges handled by the parallel merge / the sort (more: serial fallback)
enum : unsigned char { kClsNormal = 0, kClsGlobal = 1, kClsSpill = 2 };

// One workgroup.  Natural ranges (bounds[0 .. *count): their ends, ascending) are merged greedily, left to right,
// while the merged range fits the LDS budget and merge_nodes; a range whose rows do not fit is cut into spill chunks,
// one whose per-edge array does not fit stays alone ("edge-global").  Up to kPlanCache ranges this runs in parallel:
//   A  class of every range                                        (thread per range)
//   B  next[k] = where the greedy group that STARTS at k ends      (thread per range; the group condition is monotone)
//   C  the group starts: 0, next[0], next[next[0]], ...            (one thread, one LDS read per group)
//   D  exclusive scans of the fit / spill-chunk counts, output     (thread per range)
// which gives exactly the lists of the serial walk (kept below for longer lists).  Then all threads sort the fit list.
__global__ __launch_bounds__(kPlanThreads) void plan_cut_kernel(int m, int nnz, int f, int budget_bytes,
                                                                int merge_nodes,
                                                                const int *__restrict__ row_ptr, int *plan,
                                                                const PlanPair *pairs, const int *bounds,
                                                                const int *count, int coords_off) {
  extern __shared__ __attribute__((aligned(16))) int plan_lds[];
  int *s_end = plan_lds;                    // [kPlanCache + 1] prefix form: range k = [s_end[k], s_end[k + 1])
  int *s_rp = s_end + kPlanCache + 1;       // [kPlanCache + 1] row_ptr at those nodes
  int *s_bad = s_rp + kPlanCache + 1;       // [kPlanCache + 1] rows with a duplicate edge before those nodes
  int *s_next = s_bad + kPlanCache + 1;     // [kPlanCache]
  int *s_tot = s_next + kPlanCache;         // [2][kPlanThreads] scan scratch
  unsigned char *s_cls = reinterpret_cast<unsigned char *>(s_tot + 2 * kPlanThreads);  // [kPlanCache]
  unsigned char *s_start = s_cls + kPlanCache;                                          // [kPlanCache]
  __shared__ int s_nfit, s_stat[5];
  int *hdr = plan;
  int *fit = plan + kPlanHeader;
  int *spill = fit + 2 * (size_t)m;
  const int t = threadIdx.x;
  const bool dense_f = (f == 8 || f == 16 || f == 32 || f == 64 || f == 128);
  // rows with a duplicate edge before row `end`
  auto bad_before = [&](int end) { return end > 0 ? pairs[end - 1].bad : 0; };
  // a fit entry (n0, n1 | flags) for the range [n0, n1) holding ed edges; returns its flags
  auto fit_entry = [&](int slot, int n0, int n1, int ed, int bad0, int bad1) {
    const int nn = n1 - n0;
    const bool edge_global = plan_full_bytes(nn, ed, f) > budget_bytes;  // only ever true for an unmerged range
    const bool dense = dense_f && nn <= 255 && min(ed, 1 << 24) * 32 >= nn * nn && bad1 == bad0;
    const int flags = (edge_global ? kPlanEdgeGlobal : 0) | (dense ? kPlanDense : 0);
    fit[2 * slot] = n0;
    fit[2 * slot + 1] = n1 | flags;
    return flags;
  };

  const int nb = *count;
  if (nb <= kPlanCache) {
    if (t < 5) s_stat[t] = 0;
    for (int k = t; k <= nb; k += kPlanThreads) {
      const int en = k ? bounds[k - 1] : 0;
      s_end[k] = en;
      s_rp[k] = row_ptr[en];
      s_bad[k] = bad_before(en);
    }
    __syncthreads();
    for (int k = t; k < nb; k += kPlanThreads) {  // A
      const int n_one = s_end[k + 1] - s_end[k], e_one = s_rp[k + 1] - s_rp[k];
      s_cls[k] = plan_lite_bytes(n_one, e_one, f) > budget_bytes   ? kClsSpill
                 : plan_full_bytes(n_one, e_one, f) > budget_bytes ? kClsGlobal
                                                                   : kClsNormal;
      s_start[k] = 0;
    }
    __syncthreads();
    for (int k = t; k < nb; k += kPlanThreads) {  // B
      int j = k + 1;
      if (s_cls[k] == kClsNormal)
        while (j < nb && s_cls[j] == kClsNormal && s_end[j + 1] - s_end[k] <= merge_nodes &&
               plan_full_bytes(s_end[j + 1] - s_end[k], s_rp[j + 1] - s_rp[k], f) <= budget_bytes)
          ++j;
      s_next[k] = j;
    }
    __syncthreads();
    if (t == 0)  // C
      for (int k = 0; k < nb; k = s_next[k]) s_start[k] = 1;
    __syncthreads();
    // D: every group start emits one fit entry (not for a spill range) and its spill chunks; exclusive scans number them
    constexpr int PER = kPlanCache / kPlanThreads;
    int cf[PER], cs[PER], tf = 0, ts = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = t * PER + u;
      const bool st = k < nb && s_start[k];
      const bool sp = st && s_cls[k] == kClsSpill;
      cf[u] = tf;
      cs[u] = ts;
      tf += (st && !sp) ? 1 : 0;
      ts += sp ? (s_end[k + 1] - s_end[k] + kHyperRows - 1) / kHyperRows : 0;
    }
    s_tot[t] = tf;
    s_tot[kPlanThreads + t] = ts;
    __syncthreads();
    for (int o = 1; o < kPlanThreads; o <<= 1) {  // inclusive Hillis-Steele over the per-thread totals
      const int a = t >= o ? s_tot[t - o] : 0, b2 = t >= o ? s_tot[kPlanThreads + t - o] : 0;
      __syncthreads();
      s_tot[t] += a;
      s_tot[kPlanThreads + t] += b2;
      __syncthreads();
    }
    const int base_f = s_tot[t] - tf, base_s = s_tot[kPlanThreads + t] - ts;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = t * PER + u;
      if (k < nb && s_start[k]) {
        if (s_cls[k] == kClsSpill) {
          int slot = base_s + cs[u];
          for (int r = s_end[k]; r < s_end[k + 1]; r += kHyperRows, ++slot) {
            spill[2 * slot] = r;
            spill[2 * slot + 1] = min(s_end[k + 1], r + kHyperRows);
          }
        } else {
          const int j = s_next[k], ed = s_rp[j] - s_rp[k];
          const int flags = fit_entry(base_f + cf[u], s_end[k], s_end[j], ed, s_bad[k], s_bad[j]);
          atomicMax(&s_stat[0], s_end[j] - s_end[k]);
          atomicMax(&s_stat[1], ed);
          if (flags & kPlanEdgeGlobal) atomicAdd(&s_stat[2], 1);
          if (flags & kPlanDense) atomicAdd(&s_stat[3], 1);
          if ((flags & kPlanDense) && s_end[j] - s_end[k] > 128) atomicAdd(&s_stat[4], 1);
        }
      }
    }
    __syncthreads();
    if (t == 0) {
      const int nfit = s_tot[kPlanThreads - 1];
      hdr[0] = nfit;
      hdr[1] = s_tot[2 * kPlanThreads - 1];
      hdr[2] = s_stat[0];
      hdr[3] = s_stat[1];
      hdr[4] = m;
      hdr[5] = nnz;
      hdr[6] = f;
      hdr[7] = budget_bytes;
      hdr[8] = s_stat[2];
      hdr[9] = s_stat[3];
      hdr[10] = s_stat[4];  // dense ranges of more than 128 nodes: they come first in the sorted list
      hdr[11] = coords_off;
      s_nfit = nfit;
    }
  } else if (t == 0) {
    // serial walk over the global arrays (long lists of natural ranges: batches of thousands of tiny graphs)
    int nfit = 0, nspill = 0, maxn = 0, maxe = 0, nglobal = 0, ndense = 0, nwide = 0;
    auto flush = [&](int n0, int n1, int ed, int bad0, int bad1) {
      if (n1 <= n0) return;
      const int flags = fit_entry(nfit, n0, n1, ed, bad0, bad1);
      nglobal += (flags & kPlanEdgeGlobal) ? 1 : 0;
      ndense += (flags & kPlanDense) ? 1 : 0;
      nwide += ((flags & kPlanDense) && n1 - n0 > 128) ? 1 : 0;
      ++nfit;
      maxn = max(maxn, n1 - n0);
      maxe = max(maxe, ed);
    };
    int cur0 = 0, cur1 = 0, prev = 0;          // current merged range [cur0, cur1), previous range end
    int rp_cur0 = row_ptr[0], rp_cur1 = rp_cur0, rp_prev = rp_cur0;
    int bad_cur0 = 0, bad_cur1 = 0, bad_prev = 0;  // duplicate-edge row counts before cur0 / cur1 / prev
    for (int k = 0; k < nb; ++k) {
      const int end = bounds[k], rp_end = row_ptr[end], bad_end = bad_before(end);
      const int n_one = end - prev, e_one = rp_end - rp_prev;
      if (plan_lite_bytes(n_one, e_one, f) > budget_bytes) {            // not even the feature rows of this range fit
        flush(cur0, cur1, rp_cur1 - rp_cur0, bad_cur0, bad_cur1);
        for (int r = prev; r < end; r += kHyperRows) {
          spill[2 * nspill] = r;
          spill[2 * nspill + 1] = min(end, r + kHyperRows);
          ++nspill;
        }
        cur0 = cur1 = end;
        rp_cur0 = rp_cur1 = rp_end;
        bad_cur0 = bad_cur1 = bad_end;
      } else if (plan_full_bytes(n_one, e_one, f) > budget_bytes) {     // rows fit, the per-edge array goes to global scratch
        flush(cur0, cur1, rp_cur1 - rp_cur0, bad_cur0, bad_cur1);
        flush(prev, end, e_one, bad_prev, bad_end);
        cur0 = cur1 = end;
        rp_cur0 = rp_cur1 = rp_end;
        bad_cur0 = bad_cur1 = bad_end;
      } else if (cur1 > cur0 &&
                 (plan_full_bytes(end - cur0, rp_end - rp_cur0, f) > budget_bytes || end - cur0 > merge_nodes)) {
        flush(cur0, cur1, rp_cur1 - rp_cur0, bad_cur0, bad_cur1);
        cur0 = prev;
        rp_cur0 = rp_prev;
        bad_cur0 = bad_prev;
        cur1 = end;
        rp_cur1 = rp_end;
        bad_cur1 = bad_end;
      } else {
        if (cur1 == cur0) { cur0 = prev; rp_cur0 = rp_prev; bad_cur0 = bad_prev; }
        cur1 = end;
        rp_cur1 = rp_end;
        bad_cur1 = bad_end;
      }
      prev = end;
      rp_prev = rp_end;
      bad_prev = bad_end;
    }
    flush(cur0, cur1, rp_cur1 - rp_cur0, bad_cur0, bad_cur1);
    hdr[0] = nfit;
    hdr[1] = nspill;
    hdr[2] = maxn;
    hdr[3] = maxe;
    hdr[4] = m;
    hdr[5] = nnz;
    hdr[6] = f;
    hdr[7] = budget_bytes;
    hdr[8] = nglobal;
    hdr[9] = (nfit <= kPlanCache) ? ndense : 0;  // the dense ranges are only usable once sorted to the front (below)
    hdr[10] = (nfit <= kPlanCache) ? nwide : 0;
    hdr[11] = coords_off;
    s_nfit = nfit;
  }
  __syncthreads();

  // Largest ranges first (workgroups are dispatched in index order, so the tail of the launch is made of the
  // shortest ones): bitonic sort of (edges, index) in LDS, reusing s_end / s_rp.  Skipped for long lists,
  // where the tail is negligible anyway.
  const int nfit = s_nfit;
  if (nfit > 1 && nfit <= kPlanCache) {
    int N = 1;
    while (N < nfit) N <<= 1;
    for (int i = t; i < N; i += kPlanThreads) {
      if (i < nfit) {
        const int n0 = fit[2 * i], n1 = fit[2 * i + 1] & kPlanRangeMask;
        // key: dense ranges first, among them those of more than 128 nodes first (the matrix-core kernels run them
        // in a loop of their own), then by edge count
        const bool dn = (fit[2 * i + 1] & kPlanDense) != 0;
        s_end[i] = (dn ? (1 << 30) : 0) + ((dn && n1 - n0 > 128) ? (1 << 29) : 0) + min(row_ptr[n1] - row_ptr[n0], (1 << 29) - 1);
      } else {
        s_end[i] = -1;
      }
      s_rp[i] = i;
    }
    __syncthreads();
    for (int k = 2; k <= N; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = t; i < N; i += kPlanThreads) {
          const int x = i ^ j;
          if (x > i) {
            const bool desc = (i & k) == 0;
            const int a = s_end[i], c = s_end[x];
            if ((a < c) == desc && a != c) {
              s_end[i] = c; s_end[x] = a;
              const int ia = s_rp[i]; s_rp[i] = s_rp[x]; s_rp[x] = ia;
            }
          }
        }
        __syncthreads();
      }
    int *tmp = spill + 2 * (size_t)m;  // the pairs (2m ints) are not needed any more
    for (int i = t; i < nfit; i += kPlanThreads) {
      const int src = s_rp[i];
      tmp[2 * i] = fit[2 * src];
      tmp[2 * i + 1] = fit[2 * src + 1];
    }
    __threadfence_block();
    __syncthreads();
    for (int i = t; i < 2 * nfit; i += kPlanThreads) fit[i] = tmp[i];
  }
}

// coords[e] of every edge of the dense ranges (the first hdr[9] fit ranges) and the ranges' edge bitmaps: bit c of
// mask[8 i ..] <=> edge (i, n0 + c), bit r of maskT[8 j ..] <=> edge (n0 + r, j) (LDS atomics, then whole rows out).
// Runs after plan_cut_kernel on the same stream; the range count is read on the device, so no host round trip.
__global__ __launch_bounds__(256) void plan_coords_kernel(const int *__restrict__ row_ptr,
                                                          const int *__restrict__ col_ind, const int *plan,
                                                          unsigned short *__restrict__ coords,
                                                          unsigned *__restrict__ mask, unsigned *__restrict__ maskT,
                                                          unsigned short *__restrict__ ranked) {
  __shared__ unsigned s_mask[2][256 * kPlanMaskWords];
  const int ndense = plan[9];
  const int *fit = plan + kPlanHeader;
  const int grp = threadIdx.x / kExtentLanes, gl = threadIdx.x % kExtentLanes;
  for (int k = blockIdx.x; k < ndense; k += gridDim.x) {
    const int n0 = fit[2 * k], n1 = fit[2 * k + 1] & kPlanRangeMask;
    const int words = (n1 - n0) * kPlanMaskWords;
    for (int t = threadIdx.x; t < words; t += 256) s_mask[0][t] = s_mask[1][t] = 0u;
    __syncthreads();
    for (int i = n0 + grp; i < n1; i += 256 / kExtentLanes) {
      const int ea = row_ptr[i], eb = row_ptr[i + 1];
      for (int e = ea + gl; e < eb; e += kExtentLanes) {
        const int r = i - n0, c = (col_ind[e] - n0) & 0xFF;
        coords[e] = (unsigned short)((r << 8) | c);
        atomicOr(&s_mask[0][r * kPlanMaskWords + (c >> 5)], 1u << (c & 31));
        atomicOr(&s_mask[1][c * kPlanMaskWords + (r >> 5)], 1u << (r & 31));
      }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < words; t += 256) {
      mask[(size_t)n0 * kPlanMaskWords + t] = s_mask[0][t];
      maskT[(size_t)n0 * kPlanMaskWords + t] = s_mask[1][t];
      // the coordinates in rank order: word t of row r lists its set bits behind those of the row's earlier words
      const int r = t / kPlanMaskWords, w = t % kPlanMaskWords;
      int at = row_ptr[n0 + r];
      for (int v = 0; v < w; ++v) at += __popc(s_mask[0][r * kPlanMaskWords + v]);
      for (unsigned bits = s_mask[0][t]; bits; bits &= bits - 1)
        ranked[at++] = (unsigned short)((r << 8) | (32 * w + (__ffs(bits) - 1)));
    }
    __syncthreads();
  }
}

// Edge values of the dense ranges in dense form: W[256 i + c] = val[e] for the edge e from node i to the node c places
// after the first node of i's range (coords[e] holds (r, c)); the buffer was zeroed before.  One workgroup per range walk.
__global__ __launch_bounds__(256) void plan_dense_weights_kernel(const int *__restrict__ row_ptr, const int *plan,
                                                                 const unsigned short *__restrict__ coords,
                                                                 const float *__restrict__ val, float *__restrict__ W) {
  const int ndense = plan[9];
  const int *fit = plan + kPlanHeader;
  for (int k = blockIdx.x; k < ndense; k += gridDim.x) {
    const int n0 = fit[2 * k], n1 = fit[2 * k + 1] & kPlanRangeMask;
    const int ea = row_ptr[n0], eb = row_ptr[n1];
    for (int e = ea + (int)threadIdx.x; e < eb; e += 256) {
      const unsigned c = coords[e];
      W[(size_t)(n0 + (int)(c >> 8)) * kPlanWeightStride + (c & 0xFFu)] = val[e];
    }
  }
}

}  // namespace dfgnn

using namespace dfgnn;

extern "C" {

// rocPRIM temporary storage (bytes) shared by the scan and the select over m elements
static size_t plan_temp_bytes(int m) {
  size_t a = 0, b = 0;
  (void)rocprim::inclusive_scan(nullptr, a, (PlanPair *)nullptr, (PlanPair *)nullptr, (size_t)m, PlanPairAdd{},
                                (hipStream_t) nullptr);
  auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), PlanCutOpen{nullptr});
  (void)rocprim::select(nullptr, b, rocprim::counting_iterator<int>(1), flags, (int *)nullptr, (int *)nullptr, (size_t)m,
                        (hipStream_t) nullptr);
  return ((a > b ? a : b) + 255) & ~(size_t)255;
}

// scratch after the fit / spill lists: pairs (2m ints), bounds (m + 1), count (1 + 2 pad), temp (aligned to 256 B);
// then the packed edge coordinates (uint16[nnz], 16-byte aligned)
static size_t plan_coords_off(int m) {
  return (kPlanHeader + 7 * (size_t)m + 4 + (plan_temp_bytes(m) + 256) / sizeof(int) + 3) & ~(size_t)3;
}
size_t dfgnn_plan_ints(int m, int nnz) {
  if (m < 0 || nnz < 0) return 0;
  return plan_mask_off(plan_coords_off(m), nnz) + 2 * (size_t)kPlanMaskWords * (size_t)m + ((size_t)nnz + 1) / 2 + 4;
}

int dfgnn_plan_build(int m, int nnz, int f, const int *row_ptr, const int *col_ind, int *plan, int *meta_host,
                     dfgnn_stream_t stream) {
  if (m < 0 || nnz < 0 || f <= 0 || !plan || !meta_host) return kErrBadArg;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int k = 0; k < kPlanHeader; ++k) meta_host[k] = 0;
  meta_host[4] = m; meta_host[5] = nnz; meta_host[6] = f; meta_host[7] = kBlockLdsBudget;
  if (m == 0) return 0;
  if (!row_ptr || (nnz > 0 && !col_ind)) return kErrBadArg;
  PlanPair *pairs = reinterpret_cast<PlanPair *>(plan + kPlanHeader + 4 * (size_t)m);
  int *bounds = plan + kPlanHeader + 6 * (size_t)m;
  int *count = bounds + m + 1;
  void *temp = reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(count + 3) + 255) & ~(uintptr_t)255);
  size_t temp_bytes = plan_temp_bytes(m);
  if (hipError_t rc = hipMemsetAsync(pairs, 0, (size_t)m * sizeof(PlanPair), s)) return (int)rc;
  const long threads = (long)m * kExtentLanes;
  plan_row_extent_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(m, row_ptr, col_ind, pairs);
  if (int rc = launch_status()) return rc;
  size_t tb = temp_bytes;
  if (hipError_t rc = rocprim::inclusive_scan(temp, tb, pairs, pairs, (size_t)m, PlanPairAdd{}, s)) return (int)rc;
  auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), PlanCutOpen{pairs});
  tb = temp_bytes;
  if (hipError_t rc = rocprim::select(temp, tb, rocprim::counting_iterator<int>(1), flags, bounds, count, (size_t)m, s))
    return (int)rc;
  // Widths with a matrix-core form: a merged range costs n^2 there and needs a second pass over 128-row blocks past
  // 128 nodes, so small graphs are only merged up to 128 nodes.
  const int merge_nodes = (f == 8 || f == 16 || f == 32 || f == 64 || f == 128) ? 128 : kBlockMergeNodes;
  constexpr size_t kCutLds = sizeof(int) * (3 * (kPlanCache + 1) + kPlanCache + 2 * kPlanThreads) + 2 * kPlanCache;
  static_assert(kCutLds <= (size_t)kLdsBytes, "plan_cut_kernel LDS");
  if (hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void *>(plan_cut_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCutLds))
    return (int)rc;
  const size_t coords_off = plan_coords_off(m);
  if (coords_off > 0x7fffffffu) return kErrUnsupported;
  plan_cut_kernel<<<1, kPlanThreads, kCutLds, s>>>(m, nnz, f, kBlockLdsBudget, merge_nodes, row_ptr, plan, pairs, bounds,
                                                   count, (int)coords_off);
  if (int rc = launch_status()) return rc;
  if (nnz > 0) {
    unsigned *mask = reinterpret_cast<unsigned *>(plan + plan_mask_off(coords_off, nnz));
    plan_coords_kernel<<<(unsigned)min(m, 2048), 256, 0, s>>>(row_ptr, col_ind, plan,
                                                             reinterpret_cast<unsigned short *>(plan + coords_off), mask,
                                                             mask + (size_t)kPlanMaskWords * m,
                                                             reinterpret_cast<unsigned short *>(mask + 2 * (size_t)kPlanMaskWords * m));
    if (int rc = launch_status()) return rc;
  }
  if (hipError_t rc = hipMemcpyAsync(meta_host, plan, kPlanHeader * sizeof(int), hipMemcpyDeviceToHost, s)) return (int)rc;
  return (int)hipStreamSynchronize(s);
}

// Dense edge values for the WEIGHTED statistics pair (dfgnn.h): weights[256 m] floats, zero off the edges.
size_t dfgnn_plan_dense_weights_floats(int m) { return m < 0 ? 0 : (size_t)kPlanWeightStride * (size_t)m; }

int dfgnn_plan_dense_weights(int m, int nnz, const int *row_ptr, const float *val, const int *plan, const int *plan_meta,
                             float *weights, dfgnn_stream_t stream) {
  if (m < 0 || nnz < 0 || !plan || !plan_meta) return kErrBadArg;
  if (m == 0) return 0;
  if (!row_ptr || !weights || (nnz > 0 && !val)) return kErrBadArg;
  if (plan_meta[4] != m || plan_meta[5] != nnz) return kErrBadArg;  // a plan of another graph
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (hipError_t rc = hipMemsetAsync(weights, 0, dfgnn_plan_dense_weights_floats(m) * sizeof(float), s)) return (int)rc;
  if (nnz == 0 || plan_meta[9] == 0) return 0;
  const size_t coords_off = plan_coords_off(m);
  plan_dense_weights_kernel<<<(unsigned)min(plan_meta[9], 4096), 256, 0, s>>>(
      row_ptr, plan, reinterpret_cast<const unsigned short *>(plan + coords_off), val, weights);
  return launch_status();
}

}  // extern "C"
