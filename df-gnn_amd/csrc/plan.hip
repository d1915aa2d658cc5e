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

// Exclusive scan of part[0 .. kPlanThreads) in place by wave 0 (lane l owns 16 consecutive entries, a wave-level
// shuffle scan combines the lane totals).  FORWARD: part[k] <- op(identity, part[0..k)); else the mirror image
// (part[k] <- op over part(k..end)).  Every thread of the workgroup must call it (it contains the barriers).
template <bool FORWARD, class Op>
__device__ __forceinline__ void plan_scan_partials(int *part, int identity, Op op) {
  constexpr int PER = kPlanThreads / kWave;
  __syncthreads();
  if (threadIdx.x < kWave) {
    const int lane = threadIdx.x;
    int v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = part[FORWARD ? lane * PER + k : kPlanThreads - 1 - (lane * PER + k)];
    int tot = identity;
#pragma unroll
    for (int k = 0; k < PER; ++k) tot = op(tot, v[k]);
    int incl = tot;  // inclusive scan of the lane totals
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int up = __shfl_up(incl, o, kWave);
      if (lane >= o) incl = op(incl, up);
    }
    int run = __shfl_up(incl, 1, kWave);
    if (lane == 0) run = identity;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      part[FORWARD ? lane * PER + k : kPlanThreads - 1 - (lane * PER + k)] = run;
      run = op(run, v[k]);
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
  __shared__ int part[kPlanThreads];
  int *hdr = plan;
  int *fit = plan + kPlanHeader;
  int *spill = fit + 2 * (size_t)m;
  int *lo = spill + 2 * (size_t)m;
  int *hi = lo + m;
  int *bounds = hi + m;  // [m + 1] ends of the natural closed ranges
  int *unsorted = bounds + m + 1;  // [m + 1] in: per-row 'has a duplicate edge' flag, out: exclusive prefix count
  const int t = threadIdx.x;
  const int chunk = (m + kPlanThreads - 1) / kPlanThreads;
  const int b = min(m, t * chunk), e = min(m, b + chunk);

  // prefix max of hi
  int acc = -1;
  for (int i = b; i < e; ++i) acc = max(acc, hi[i]);
  part[t] = acc;
  plan_scan_partials<true>(part, -1, [](int a, int b) { return max(a, b); });
  acc = part[t];
  for (int i = b; i < e; ++i) { acc = max(acc, hi[i]); hi[i] = acc; }
  __syncthreads();
  // suffix min of lo
  acc = m;
  for (int i = e - 1; i >= b; --i) acc = min(acc, lo[i]);
  part[t] = acc;
  plan_scan_partials<false>(part, m, [](int a, int b) { return min(a, b); });
  acc = part[t];
  for (int i = e - 1; i >= b; --i) { acc = min(acc, lo[i]); lo[i] = acc; }
  __syncthreads();
  // exclusive prefix count of the rows with duplicate edges (unsorted[m] = total)
  {
    int c = 0;
    for (int i = b; i < e; ++i) c += unsorted[i];
    part[t] = c;
    plan_scan_partials<true>(part, 0, [](int a, int b) { return a + b; });
    int run = part[t];
    for (int i = b; i < e; ++i) {
      const int v = unsorted[i];
      unsorted[i] = run;
      run += v;
    }
    if (e == m) unsorted[m] = run;  // every thread whose chunk ends at m holds the total
    __syncthreads();
  }
  // count boundaries per thread chunk, scan, write
  int cnt = 0;
  for (int i = b; i < e; ++i) cnt += (hi[i] <= i && (i + 1 == m || lo[i + 1] >= i + 1)) ? 1 : 0;
  part[t] = cnt;
  plan_scan_partials<true>(part, 0, [](int a, int b) { return a + b; });
  if (t == kPlanThreads - 1) hdr[7] = part[t] + cnt;  // number of natural ranges (temporary)
  __syncthreads();
  int pos = part[t];
  for (int i = b; i < e; ++i)
    if (hi[i] <= i && (i + 1 == m || lo[i + 1] >= i + 1)) bounds[pos++] = i + 1;
  __syncthreads();

  // Greedy merge by thread 0.  The ends of the natural ranges and their row_ptr values are first copied to
  // LDS (when there are few enough) so the serial walk does not pay a global-memory round trip per range.
  constexpr int kCache = 4096;
  __shared__ int s_end[kCache], s_rp[kCache];
  __shared__ int s_nfit;
  const int nb = hdr[7];
  const bool cached = nb <= kCache;
  if (cached)
    for (int k = t; k < nb; k += kPlanThreads) {
      const int en = bounds[k];
      s_end[k] = en;
      s_rp[k] = row_ptr[en];
    }
  __syncthreads();
  if (t == 0) {
    int nfit = 0, nspill = 0, maxn = 0, maxe = 0, nglobal = 0, ndense = 0;
    const bool dense_f = (f == 32 || f == 64 || f == 128);
    // LDS bytes of a range [n0, n1) holding ed edges: resident rows + 1/sum + rebased row_ptr + narrowed
    // column ids, plus (full only) the per-edge fp32 array.  Layout: dfgnn_block.hpp:carve_block_lds.
    auto lite = [&](long n, long ed) -> long { return n * (long)f * 4 + n * 8 + ed * (n <= 256 ? 1 : 2); };
    auto full = [&](long n, long ed) -> long { return lite(n, ed) + 4 * ed; };
    auto flush = [&](int n0, int n1, int ed) {
      if (n1 <= n0) return;
      const bool edge_global = full(n1 - n0, ed) > budget_bytes;  // only ever true for an unmerged range
      const long nn = n1 - n0;
      const bool dense = dense_f && nn <= 255 && (long)ed * 32 >= nn * nn && unsorted[n1] == unsorted[n0];
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
    for (int k = 0; k < nb; ++k) {
      const int end = cached ? s_end[k] : bounds[k];
      const int rp_end = cached ? s_rp[k] : row_ptr[end];
      const long n_one = end - prev, e_one = rp_end - rp_prev;
      if (lite(n_one, e_one) > budget_bytes) {            // not even the feature rows of this range fit
        flush(cur0, cur1, rp_cur1 - rp_cur0);
        for (int r = prev; r < end; r += kHyperRows) {
          spill[2 * nspill] = r;
          spill[2 * nspill + 1] = min(end, r + kHyperRows);
          ++nspill;
        }
        cur0 = cur1 = end;
        rp_cur0 = rp_cur1 = rp_end;
      } else if (full(n_one, e_one) > budget_bytes) {     // rows fit, the per-edge array goes to global scratch
        flush(cur0, cur1, rp_cur1 - rp_cur0);
        flush(prev, end, (int)e_one);
        cur0 = cur1 = end;
        rp_cur0 = rp_cur1 = rp_end;
      } else if (cur1 > cur0 && (full(end - cur0, rp_end - rp_cur0) > budget_bytes || end - cur0 > merge_nodes)) {
        flush(cur0, cur1, rp_cur1 - rp_cur0);
        cur0 = prev;
        rp_cur0 = rp_prev;
        cur1 = end;
        rp_cur1 = rp_end;
      } else {
        if (cur1 == cur0) { cur0 = prev; rp_cur0 = rp_prev; }
        cur1 = end;
        rp_cur1 = rp_end;
      }
      prev = end;
      rp_prev = rp_end;
    }
    flush(cur0, cur1, rp_cur1 - rp_cur0);
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
