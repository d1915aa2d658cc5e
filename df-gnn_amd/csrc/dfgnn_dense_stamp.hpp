// dfgnn_dense_stamp.hpp -- phase stamps / workgroup trace of the matrix-core kernels (diagnostic builds only:
// -DDFGNN_STAMPS, and only in the translation unit that defines DFGNN_STAMPS_TU -- gt_dense.hip -- which owns the two
// device variables).  Everywhere else the macros are empty.
#pragma once
#include "dfgnn_dense.hpp"

namespace dfgnn {

#if defined(DFGNN_STAMPS) && !defined(DFGNN_STAMPS_TU)
#undef DFGNN_STAMPS
#endif
#ifdef DFGNN_STAMPS
__device__ unsigned long long *dfgnn_dense_stamps = nullptr;  // [wg][16] phase boundaries (diagnostic build only)
#define DFGNN_DSTAMP(k)                                                                             \
  if (threadIdx.x == 0 && dfgnn_dense_stamps)                                                       \
    dfgnn_dense_stamps[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memtime();
// per-workgroup trace (same diagnostic build): [wg][8] = entry time, exit time (s_memtime), hardware id
// (XCC_ID << 32 | HW_ID: which CU ran it), entry and exit time on the constant 100 MHz clock
__device__ unsigned long long *dfgnn_wg_trace = nullptr;
#define DFGNN_TRACE_IN                                                                              \
  if (threadIdx.x == 0 && dfgnn_wg_trace) {                                                         \
    unsigned long long *t_ = dfgnn_wg_trace + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8;   \
    t_[0] = __builtin_amdgcn_s_memtime();                                                           \
    t_[3] = __builtin_amdgcn_s_memrealtime();                                                       \
    t_[2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4); \
  }
#define DFGNN_TRACE_OUT                                                                             \
  if (threadIdx.x == 0 && dfgnn_wg_trace) {                                                         \
    unsigned long long *t_ = dfgnn_wg_trace + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 8;   \
    t_[1] = __builtin_amdgcn_s_memtime();                                                           \
    t_[4] = __builtin_amdgcn_s_memrealtime();                                                       \
  }
#else
#define DFGNN_DSTAMP(k)
#define DFGNN_TRACE_IN
#define DFGNN_TRACE_OUT
#endif

}  // namespace dfgnn

#ifdef DFGNN_STAMPS  // (only in the translation unit that owns the variables: see the top of this file)
extern "C" int dfgnn_debug_set_dense_stamps(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_dense_stamps), &p, sizeof(p));
}
extern "C" int dfgnn_debug_set_wg_trace(void *p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(dfgnn::dfgnn_wg_trace), &p, sizeof(p));
}
#endif
