/* dfgnn_errstr.h -- texts of the argument errors of the C ABI (include/dfgnn.h: DFGNN_E_*), shared by the kernels
 * library (capi.hip) and the forwarder (gen_shim.py), which answers them without loading the kernels. */
#pragma once
static inline const char *dfgnn_static_error_string(int code) { /* NULL: a hipError_t, ask the HIP runtime */
  if (code == 0) return "success";
  if (code == -1) return "dfgnn: bad argument (negative size or NULL required pointer)";
  if (code == -2)
    return "dfgnn: unsupported shape (f > 1024, f % 4 != 0 with f > 256, or h > 65535) or, for the statistics-saving pair, a "
           "batch that the matrix-core kernels do not cover (dfgnn_gt_stats_applies)";
  if (code < 0) return "dfgnn: unknown error";
  return 0;
}
