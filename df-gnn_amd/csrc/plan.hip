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
//   [12+4m .. )         scratch: lo[m], hi[m], bounds[m+1], unsorted[m+1]
//
// "Dense" (gt_dense.hip, dfgnn_dense.hpp): at most 255 nodes, at least one edge per 32 node pairs, f in {32, 64, 128}
// and no duplicate edge in any row of the range -- the one thing a dense mask cannot represent.
#include <type_traits>

#include "../../include/dfgnn.h"
#include "dfgnn_launch.hpp"

namespace dfgnn {

constexpr int kPlanThreads = 1024;

__global__ void plan_row_extent_kernel(int m, const int *__restrict__ row_ptr, const int *__restrict__ col_ind,
                                       int *__restrict__ lo, int *__restrict__ hi, int *__restrict__ unsorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int ea = row_ptr[i], eb = row_ptr[i + 1];
  int cl = i, ch = i;
  for (int e = ea; e < eb; ++e) {
    const int c = col_ind[e];
    cl = min(cl, c);
    ch = max(ch, c);
  }
  lo[i] = cl;
  hi[i] = ch;
  // Duplicate columns in the row?  Only rows that could sit in a dense range matter (the range holds cl .. ch, so
  // its span and its length are below 256): those are checked against a 256-bit map of the columns seen so far.
  int bad = 1;
  if (ch - cl < 256 && eb - ea < 256) {
    unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    bad = 0;
    for (int e = ea; e < eb; ++e) {
      const int k = col_ind[e] - cl;
      const unsigned long long bit = 1ull << (k & 63);
      const int q = k >> 6;
      const unsigned long long cur = q == 0 ? w0 : q == 1 ? w1 : q == 2 ? w2 : w3;
      bad |= (cur & bit) ? 1 : 0;
      w0 |= q == 0 ? bit : 0;
      w1 |= q == 1 ? bit : 0;
      w2 |= q == 2 ? bit : 0;
      w3 |= q == 3 ? bit : 0;
    }
  }
  unsorted[i] = bad;  // 1: the row has a duplicate edge (or is too wide / long to be part of a dense range)
}

// In-place scan over m elements by the whole workgroup with coalesced accesses: wave w owns a contiguous span
// (a multiple of 64 long), its lanes step through it 64 elements at a time (FORWARD: ascending; else descending, i.e.
// a suffix scan); a wave-level shuffle scan plus a running carry gives every element its inclusive and exclusive
// value: store(i, inclusive, exclusive).  load(i) may compute the element (fused flag evaluation).
// Every thread must call it; it ends with a __syncthreads() (global writes of the pass are visible to the next).
template <bool FORWARD, class Op, class Load, class Store>
__device__ __forceinline__ void plan_block_scan(int m, int identity, int *wave_tot, Op op, Load load, Store store) {
  constexpr int kWaves = kPlanThreads / kWave;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int span = ((m + kWaves * kWave - 1) / (kWaves * kWave)) * kWave;
  const int r0 = min(m, w * span), r1 = min(m, r0 + span);
  auto index = [&](int k) { return FORWARD ? r0 + k : r1 - 1 - k; };  // k-th element of this wave's span, scan order
  constexpr int UNR = 8;  // elements per lane in flight: the loop is bound by the latency of its loads
  int tot = identity;
  for (int base = 0; base < r1 - r0; base += UNR * kWave) {
    int x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = base + u * kWave + lane;
      x[u] = (k < r1 - r0) ? load(index(k)) : identity;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) tot = op(tot, x[u]);
  }
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) tot = op(tot, __shfl_xor(tot, o, kWave));
  if (lane == 0) wave_tot[w] = tot;
  __syncthreads();
  int carry = identity;
  for (int k = 0; k < kWaves; ++k)
    if (FORWARD ? k < w : k > w) carry = op(carry, wave_tot[k]);
  for (int base = 0; base < r1 - r0; base += UNR * kWave) {
    int x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = base + u * kWave + lane;
      x[u] = (k < r1 - r0) ? load(index(k)) : identity;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int k = base + u * kWave + lane;
      int v = x[u];
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const int up = __shfl_up(v, o, kWave);
        if (lane >= o) v = op(v, up);
      }
      const int incl = op(carry, v);
      int excl = __shfl_up(incl, 1, kWave);
      if (lane == 0) excl = carry;
      if (k < r1 - r0) store(index(k), incl, excl);
      carry = __shfl(incl, kWave - 1, kWave);
    }
  }
  __syncthreads();
}

// One workgroup.  (1) hi <- inclusive prefix max, lo <- inclusive suffix min.  (2) boundary after row i
// iff pmax[i] <= i and smin[i+1] >= i+1.  (3) thread 0 merges consecutive closed ranges greedily while
// they fit the LDS budget and emits fit blocks / spill chunks.
__global__ __launch_bounds__(kPlanThreads) void plan_cut_kernel(int m, int nnz, int f, int budget_bytes,
                                                                int merge_nodes,
                                                                const int *__restrict__ row_ptr, int *plan) {
  int *hdr = plan;
  int *fit = plan + kPlanHeader;
  int *spill = fit + 2 * (size_t)m;
  int *lo = spill + 2 * (size_t)m;
  int *hi = lo + m;
  int *bounds = hi + m;  // [m + 1] ends of the natural closed ranges
  int *unsorted = bounds + m + 1;  // [m + 1] in: per-row 'has a duplicate edge' flag, out: exclusive prefix count
  const int t = threadIdx.x;
  __shared__ int wave_tot[kPlanThreads / kWave];

  // (1) prefix max of hi, suffix min of lo, exclusive prefix count of the rows with duplicate edges
  plan_block_scan<true>(m, -1, wave_tot, [](int a, int b) { return max(a, b); }, [&](int i) { return hi[i]; },
                        [&](int i, int incl, int) { hi[i] = incl; });
  plan_block_scan<false>(m, m, wave_tot, [](int a, int b) { return min(a, b); }, [&](int i) { return lo[i]; },
                         [&](int i, int incl, int) { lo[i] = incl; });
  plan_block_scan<true>(m, 0, wave_tot, [](int a, int b) { return a + b; }, [&](int i) { return unsorted[i]; },
                        [&](int i, int incl, int excl) {
                          unsorted[i] = excl;
                          if (i == m - 1) unsorted[m] = incl;
                        });
  // (2) a boundary after row i iff pmax[i] <= i and smin[i+1] >= i+1: number them, write the ends of the natural ranges
  plan_block_scan<true>(m, 0, wave_tot, [](int a, int b) { return a + b; },
                        [&](int i) { return (hi[i] <= i && (i + 1 == m || lo[i + 1] >= i + 1)) ? 1 : 0; },
                        [&](int i, int incl, int excl) {
                          if (incl != excl) bounds[excl] = i + 1;
                          if (i == m - 1) hdr[7] = incl;  // number of natural ranges (temporary)
                        });

  // Greedy merge by thread 0.  The ends of the natural ranges and their row_ptr values are first copied to
  // LDS (when there are few enough) so the serial walk does not pay a global-memory round trip per range.
  constexpr int kCache = 4096;
  __shared__ int s_end[kCache], s_rp[kCache], s_bad[kCache];
  __shared__ int s_nfit;
  const int nb = hdr[7];
  const bool cached = nb <= kCache;
  if (cached)
    for (int k = t; k < nb; k += kPlanThreads) {
      const int en = bounds[k];
      s_end[k] = en;
      s_rp[k] = row_ptr[en];
      s_bad[k] = unsorted[en];  // rows with duplicate edges before the end of natural range k
    }
  __syncthreads();
  if (t == 0) {
    int nfit = 0, nspill = 0, maxn = 0, maxe = 0, nglobal = 0, ndense = 0;
    const bool dense_f = (f == 32 || f == 64 || f == 128);
    // LDS bytes of a range [n0, n1) holding ed edges: resident rows + 1/sum + rebased row_ptr + narrowed
    // column ids, plus (full only) the per-edge fp32 array.  Layout: dfgnn_block.hpp:carve_block_lds.
    // (32-bit arithmetic on clamped sizes -- this serial walk is latency-bound, and anything clamped is far over the
    // budget anyway)
    auto lite = [&](int n, int ed) -> int {
      n = min(n, 1 << 16);
      ed = min(ed, 1 << 24);
      return n * (4 * f + 8) + ed * (n <= 256 ? 1 : 2);
    };
    auto full = [&](int n, int ed) -> int { return lite(n, ed) + 4 * min(ed, 1 << 24); };
    auto flush = [&](int n0, int n1, int ed, int bad0, int bad1) {
      if (n1 <= n0) return;
      const bool edge_global = full(n1 - n0, ed) > budget_bytes;  // only ever true for an unmerged range
      const int nn = n1 - n0;
      const bool dense = dense_f && nn <= 255 && min(ed, 1 << 24) * 32 >= nn * nn && bad1 == bad0;
      fit[2 * nfit] = n0;
      fit[2 * nfit + 1] = n1 | (edge_global ? kPlanEdgeGlobal : 0) | (dense ? kPlanDense : 0);
      nglobal += edge_global ? 1 : 0;
      ndense += dense ? 1 : 0;
      ++nfit;
      maxn = max(maxn, n1 - n0);
      maxe = max(maxe, ed);
    };
    int cur0 = 0, cur1 = 0, prev = 0;          // current merged range [cur0, cur1), previous range end
    int rp_cur0 = row_ptr[0], rp_cur1 = rp_cur0, rp_prev = rp_cur0;
    int bad_cur0 = 0, bad_cur1 = 0, bad_prev = 0;  // duplicate-edge row counts before cur0 / cur1 / prev
    // (two instances of the walk: with a `cached ? lds : global` select per value the compiler issues the dependent
    // global loads unconditionally -- 900 cycles per range instead of 60)
    auto walk = [&](auto cached_c) {
    constexpr bool kCached = decltype(cached_c)::value;
    for (int k = 0; k < nb; ++k) {
      int end, rp_end, bad_end;
      if constexpr (kCached) {
        end = s_end[k];
        rp_end = s_rp[k];
        bad_end = s_bad[k];
      } else {
        end = bounds[k];
        rp_end = row_ptr[end];
        bad_end = unsorted[end];
      }
      const int n_one = end - prev, e_one = rp_end - rp_prev;
      if (lite(n_one, e_one) > budget_bytes) {            // not even the feature rows of this range fit
        flush(cur0, cur1, rp_cur1 - rp_cur0, bad_cur0, bad_cur1);
        for (int r = prev; r < end; r += kHyperRows) {
          spill[2 * nspill] = r;
          spill[2 * nspill + 1] = min(end, r + kHyperRows);
          ++nspill;
        }
        cur0 = cur1 = end;
        rp_cur0 = rp_cur1 = rp_end;
        bad_cur0 = bad_cur1 = bad_end;
      } else if (full(n_one, e_one) > budget_bytes) {     // rows fit, the per-edge array goes to global scratch
        flush(cur0, cur1, rp_cur1 - rp_cur0, bad_cur0, bad_cur1);
        flush(prev, end, e_one, bad_prev, bad_end);
        cur0 = cur1 = end;
        rp_cur0 = rp_cur1 = rp_end;
        bad_cur0 = bad_cur1 = bad_end;
      } else if (cur1 > cur0 && (full(end - cur0, rp_end - rp_cur0) > budget_bytes || end - cur0 > merge_nodes)) {
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
    };
    if (cached) walk(std::true_type{});
    else walk(std::false_type{});
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
    hdr[9] = (nfit <= kCache) ? ndense : 0;  // the dense ranges are only usable once sorted to the front (below)
    hdr[10] = hdr[11] = 0;
    s_nfit = nfit;
  }
  __syncthreads();

  // Largest ranges first (workgroups are dispatched in index order, so the tail of the launch is made of the
  // shortest ones): bitonic sort of (edges, index) in LDS, reusing s_end / s_rp.  Skipped for long lists,
  // where the tail is negligible anyway.
  const int nfit = s_nfit;
  if (nfit > 1 && nfit <= kCache) {
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
    int *tmp = lo;  // lo[] and hi[] (2m ints) are free now
    for (int i = t; i < nfit; i += kPlanThreads) {
      const int src = s_rp[i];
      tmp[2 * i] = fit[2 * src];
      tmp[2 * i + 1] = fit[2 * src + 1];
    }
    __syncthreads();
    for (int i = t; i < 2 * nfit; i += kPlanThreads) fit[i] = tmp[i];
  }
}

}  // namespace dfgnn

using namespace dfgnn;

extern "C" {

size_t dfgnn_plan_ints(int m) { return m < 0 ? 0 : kPlanHeader + 8 * (size_t)m + 2; }

int dfgnn_plan_build(int m, int nnz, int f, const int *row_ptr, const int *col_ind, int *plan, int *meta_host,
                     dfgnn_stream_t stream) {
  if (m < 0 || nnz < 0 || f <= 0 || !plan || !meta_host) return kErrBadArg;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int k = 0; k < kPlanHeader; ++k) meta_host[k] = 0;
  meta_host[4] = m; meta_host[5] = nnz; meta_host[6] = f; meta_host[7] = kBlockLdsBudget;
  if (m == 0) return 0;
  if (!row_ptr || (nnz > 0 && !col_ind)) return kErrBadArg;
  int *lo = plan + kPlanHeader + 4 * (size_t)m;
  int *hi = lo + m;
  int *unsorted = hi + m + (m + 1);
  plan_row_extent_kernel<<<(m + 255) / 256, 256, 0, s>>>(m, row_ptr, col_ind, lo, hi, unsorted);
  if (int rc = launch_status()) return rc;
  // Widths with a matrix-core form: a merged range costs n^2 there and needs a second pass over 128-row blocks past
  // 128 nodes, so small graphs are only merged up to 128 nodes.
  const int merge_nodes = (f == 32 || f == 64 || f == 128) ? 128 : kBlockMergeNodes;
  plan_cut_kernel<<<1, kPlanThreads, 0, s>>>(m, nnz, f, kBlockLdsBudget, merge_nodes, row_ptr, plan);
  if (int rc = launch_status()) return rc;
  if (hipError_t rc = hipMemcpyAsync(meta_host, plan, kPlanHeader * sizeof(int), hipMemcpyDeviceToHost, s)) return (int)rc;
  return (int)hipStreamSynchronize(s);
}

}  // extern "C"
