// preprocess.hip -- GPU-side graph preprocessing: COO edge list -> the 'hyper' (CSR + sorted COO rows) and CSC
// arrays the fused operators take.
//
// Replaces, for graphs that live on the GPU, the DGL-sparse calls of the reference's preprocess_* functions
// (DFGNN/layers/util.py:52-57 g_to_SPmatrix, :82-100 preprocess_Hyper, :116-142 preprocess_Hyper_fw_bw:
// `A.csr()`, `torch.sort(A.row)`, `dglsp.from_csr(...).csc()`), which the reference counts as a first-class cost of
// a training epoch (DFGNN/script/train/train_batch_graph_timing.py:115-143, 237-245).  Same results, same order:
//   CSR  = stable sort of the edge list by row   (edge_order[s] = COO position of CSR slot s, A.val[edge_order] = val)
//   CSC  = stable sort of the CSR list by column (val_idx[t]   = CSR slot of CSC entry t)
// Two LSD radix sorts (rocPRIM, int32 keys, only the ceil(log2 m) low bits -- 3 passes for a PATTERN batch where a
// generic 64-bit sort takes 8+) on keys narrowed on the fly from the int64 ids DGL hands out, plus three small
// kernels; no host synchronisation, no allocation (caller workspace).
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/dfgnn.h"
#include "dfgnn_launch.hpp"

namespace dfgnn {

// node ids as the graph library stores them (int64 for DGL's default idtype, else int32), narrowed and clamped to
// [0, m): an out-of-range id cannot make the kernels below write out of bounds
struct IdNarrow {
  const void *p;
  int wide, m;
  __host__ __device__ int operator()(int e) const {
    const long long v = wide ? static_cast<const long long *>(p)[e] : static_cast<const int *>(p)[e];
    return (int)(v < 0 ? 0 : (v >= m ? m - 1 : v));
  }
};

// ptr[r] = first position s with keys[s] >= r, for r in [0, m] (keys sorted ascending): one binary search per
// pointer entry -- uniform cost whatever the degree distribution (a scatter from the key boundaries would leave one
// thread filling the whole pointer array of a graph with many isolated nodes).
__global__ __launch_bounds__(256) void csr_pointers_kernel(int m, int nnz, const int *__restrict__ keys,
                                                           int *__restrict__ ptr) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > m) return;
  int lo = 0, hi = nnz;
  while (lo < hi) {
    const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
    if (keys[mid] < r) lo = mid + 1;
    else hi = mid;
  }
  ptr[r] = lo;
}

// out[s] = narrow(ids[order[s]])
__global__ __launch_bounds__(256) void gather_ids_kernel(int nnz, IdNarrow ids, const int *__restrict__ order,
                                                         int *__restrict__ out) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nnz) out[s] = ids(order[s]);
}

__global__ __launch_bounds__(256) void gather_int_kernel(int nnz, const int *__restrict__ src,
                                                         const int *__restrict__ order, int *__restrict__ out) {
  const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < nnz) out[s] = src[order[s]];
}

static inline unsigned key_bits(int m) {
  unsigned b = 1;
  while (b < 31 && (1u << b) < (unsigned)m) ++b;
  return b;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Keys of up to 18 bits (graphs of up to 262 144 nodes -- every batched workload) are sorted in TWO onesweep passes of
// 9 bits instead of the three 8-bit passes of rocPRIM's default configuration.
using Sort9 = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                         rocprim::radix_sort_onesweep_config<
                                             rocprim::kernel_config<256, 12>, rocprim::kernel_config<256, 16>, 9,
                                             rocprim::block_radix_rank_algorithm::match>>;
constexpr unsigned kSort9MaxBits = 18;

template <class KeysIn, class ValsIn>
static hipError_t sort_pairs(void *temp, size_t &bytes, KeysIn keys_in, int *keys_out, ValsIn vals_in, int *vals_out,
                             unsigned n, unsigned bits, hipStream_t s) {
  if (bits <= kSort9MaxBits)
    return rocprim::radix_sort_pairs<Sort9>(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, bits, s);
  return rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, 0, bits, s);
}

// rocPRIM temporary storage of the larger of the two sorts
static int sort_temp_bytes(int m, int nnz, size_t &bytes) {
  bytes = 0;
  size_t a = 0, b = 0;
  const IdNarrow ids{nullptr, 1, m};
  auto keys_a = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), ids);
  if (hipError_t rc = sort_pairs(nullptr, a, keys_a, (int *)nullptr, rocprim::counting_iterator<int>(0), (int *)nullptr,
                                 (unsigned)nnz, key_bits(m), (hipStream_t) nullptr))
    return (int)rc;
  if (hipError_t rc = sort_pairs(nullptr, b, (const int *)nullptr, (int *)nullptr, rocprim::counting_iterator<int>(0),
                                 (int *)nullptr, (unsigned)nnz, key_bits(m), (hipStream_t) nullptr))
    return (int)rc;
  bytes = align256(a > b ? a : b);
  return 0;
}

}  // namespace dfgnn

using namespace dfgnn;

extern "C" {

size_t dfgnn_preprocess_ws_bytes(int m, int nnz) {
  if (m <= 0 || nnz <= 0) return 256;
  size_t temp = 0;
  if (sort_temp_bytes(m, nnz, temp)) return 0;
  return temp + align256((size_t)nnz * sizeof(int));  // + the sorted column keys of the CSC sort
}

int dfgnn_preprocess_hyper(int m, int nnz, const void *src, const void *dst, int idx64, int *row_ptr, int *col_ind,
                           int *rows, int *edge_order, int *col_ptr, int *row_ind, int *val_idx, void *ws,
                           size_t ws_bytes, dfgnn_stream_t stream) {
  if (m < 0 || nnz < 0) return kErrBadArg;
  if (!row_ptr) return kErrBadArg;
  const bool want_csc = col_ptr || row_ind || val_idx;
  if (want_csc && !col_ptr) return kErrBadArg;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (nnz == 0) {  // pointers only
    if (hipError_t rc = hipMemsetAsync(row_ptr, 0, ((size_t)m + 1) * sizeof(int), s)) return (int)rc;
    if (want_csc)
      if (hipError_t rc = hipMemsetAsync(col_ptr, 0, ((size_t)m + 1) * sizeof(int), s)) return (int)rc;
    return 0;
  }
  if (m == 0) return kErrBadArg;  // edges without nodes
  if (!src || !dst || !col_ind || !rows || !edge_order || !ws) return kErrBadArg;
  if (want_csc && (!row_ind || !val_idx)) return kErrBadArg;
  size_t temp = 0;
  if (int rc = sort_temp_bytes(m, nnz, temp)) return rc;
  if (ws_bytes < temp + align256((size_t)nnz * sizeof(int))) return kErrBadArg;
  int *sorted_cols = reinterpret_cast<int *>(static_cast<char *>(ws) + temp);
  const unsigned bits = key_bits(m);
  const int eb = (nnz + 255) / 256, pb = m / 256 + 1;  // pointer kernel: m + 1 threads

  // CSR: (row, COO position) sorted by row; the sorted keys ARE the `rows` array of the hyper format
  const IdNarrow srcs{src, idx64, m}, dsts{dst, idx64, m};
  auto row_keys = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), srcs);
  size_t tb = temp;
  if (hipError_t rc = sort_pairs(ws, tb, row_keys, rows, rocprim::counting_iterator<int>(0), edge_order, (unsigned)nnz,
                                 bits, s))
    return (int)rc;
  csr_pointers_kernel<<<pb, 256, 0, s>>>(m, nnz, rows, row_ptr);
  gather_ids_kernel<<<eb, 256, 0, s>>>(nnz, dsts, edge_order, col_ind);
  if (int rc = launch_status()) return rc;
  if (!want_csc) return 0;

  // CSC: (column, CSR slot) sorted by column
  tb = temp;
  if (hipError_t rc = sort_pairs(ws, tb, (const int *)col_ind, sorted_cols, rocprim::counting_iterator<int>(0), val_idx,
                                 (unsigned)nnz, bits, s))
    return (int)rc;
  csr_pointers_kernel<<<pb, 256, 0, s>>>(m, nnz, sorted_cols, col_ptr);
  gather_int_kernel<<<eb, 256, 0, s>>>(nnz, rows, val_idx, row_ind);
  return launch_status();
}

}  // extern "C"
