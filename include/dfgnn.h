/*
 * dfgnn.h -- C ABI of the MI355X (gfx950) fused attention-GNN convolution library (libdfgnn.so).
 *
 * One entry point per live function of the reference's two extension modules
 * (`fused_gtconv`, `fused_gatconv`).  Plain pointers and sizes only: no torch types, no
 * allocation, no host synchronisation inside (every call is graph-capturable); the caller
 * owns all buffers (device memory) and passes the HIP stream to launch on.
 *
 * Common conventions (reference: DFGNN/src/fused_gtconv/fused_gtconv_hyper.cu:679-691)
 *   m        number of nodes  (= row_ptr length - 1)
 *   nnz      number of edges  (= col_ind length)
 *   h, f     heads, per-head feature width; features are fp32 [m, h, f] row-major, contiguous
 *   row_ptr  int32[m+1]  CSR row pointers        col_ind int32[nnz] CSR column ids
 *   rows     int32[nnz]  sorted COO row ids (the COO half of the "hyper" CSR+COO format:
 *                        rows[e] = row of CSR edge e); required wherever it appears
 *   val      fp32[nnz]   edge values multiplied into the GT logits; NULL means all ones
 *   attn_edge / grad_edge  fp32[h, nnz], head-major, CSR edge order
 *   stream   hipStream_t (passed as void*), NULL = the legacy default stream
 *
 * Return value: 0 on success; a positive value is the hipError_t of the failed launch; a
 * negative value is one of the DFGNN_E_* argument errors below.  dfgnn_error_string() decodes
 * both.  Zero-sized problems (m == 0) succeed without launching.
 *
 * All functions compute, per head, for every row i with CSR neighbours j (duplicates kept):
 *   P_e = softmax_j(s_e),  out[i,h,:] = sum_e P_e * V[j,h,:],  empty row -> 0
 * with s_e = val_e * <Q[i,h,:],K[j,h,:]> (GT) or LeakyReLU(attn_row[i,h] + attn_col[j,h]) (GAT).
 */
#ifndef DFGNN_H_
#define DFGNN_H_

#ifdef __cplusplus
extern "C" {
#endif

#define DFGNN_ABI_VERSION 11

#define DFGNN_E_BADARG (-1)      /* negative size / NULL required pointer                      */
#define DFGNN_E_UNSUPPORTED (-2) /* feature width outside the compiled range (f > 1024, or     */
                                 /* f % 4 != 0 with f > 256)                                   */

typedef void *dfgnn_stream_t; /* hipStream_t */

int dfgnn_abi_version(void);
const char *dfgnn_error_string(int code);
/* 16 hex digits: sha256 of the library's sources (csrc Makefile list, in that order) at build time -- lets a
 * caller detect a stale libdfgnn.so whose ABI number still matches. */
const char *dfgnn_build_id(void);

/* ---- block plan (optional, MI355X-specific; no counterpart in the reference) --------------------
 * A batch of small graphs (DGL GraphDataLoader, DFGNN/script/test/test_batch_graph.py:67-71) is a
 * block-diagonal adjacency: each member graph is a contiguous node range whose edges stay inside it.
 * dfgnn_plan_build finds those closed ranges on the GPU and marks the ones whose feature rows
 * (f floats per node and head) plus per-edge scratch fit one CU's 160 KB LDS; the 'hyper' entry points
 * that accept a plan then run one workgroup per range with K/V (GAT: X) resident in LDS, and cut
 * everything else into 16-row chunks for the general kernels.  Results are identical with or
 * without a plan.  The plan depends on the graph structure and on f only; build it once per batch
 * (it belongs to preprocessing, like the reference's preprocess_Hyper, DFGNN/layers/util.py:82-100).
 * Ranges that are dense (>= 1 edge per 32 node pairs), have at most 255 nodes, f in {8, 16, 32, 64, 128} and no
 * duplicate edges are additionally marked for the matrix-core kernels, which the GT
 * forward / backward use for them when val == NULL (unit edge values): masked dense attention on MFMA with
 * fp32-equivalent arithmetic (operands as fp16 hi + lo halves under power-of-two scales, fp32 accumulation:
 * ~3 x 2^-24 relative error per product, that of an fp32 FMA chain).
 *   plan       device buffer of dfgnn_plan_ints(m, nnz) int32 (lists of ranges, build scratch and, for the
 *              matrix-core kernels, 2 bytes per edge -- its row and column within its dense range -- and two edge
 *              bitmaps of 32 bytes per node: the out- and the in-neighbours of a node within its dense range)
 *   meta_host  host buffer of 12 int32 filled on return: num_fit, num_spill, max_fit_nodes,
 *              max_fit_edges, m, nnz, f, lds_budget, num_edge_global, num_dense, num_dense_wide (dense ranges of
 *              more than 128 nodes), coords_offset (int32 offset of the per-edge coordinates in `plan`)
 * dfgnn_plan_build synchronises `stream` (it copies the 12 header words back); nothing else in this
 * library does. */
size_t dfgnn_plan_ints(int m, int nnz);
/* 1 if the entry points below would use a plan with this host header for (m, nnz, h, f), else 0 (they then take the
 * general kernels, same results): the header must have been built for the same m, nnz, f; graphs with fewer than 8 edges
 * per row on average and feature matrices of 4 GiB or more (m h f 4 >= 2^32: the plan kernels address a feature row
 * with 32-bit byte offsets) do not use it.  Host-only, no GPU call. */
int dfgnn_plan_applies(int m, int nnz, int h, int f, const int *plan_meta);
int dfgnn_plan_build(int m, int nnz, int f, const int *row_ptr, const int *col_ind, int *plan,
                     int *meta_host, dfgnn_stream_t stream);

/* ---- graph preprocessing on the GPU -----------------------------------------------------------------
 * COO edge list -> the arrays of the reference's preprocess_Hyper / preprocess_Hyper_fw_bw
 * (DFGNN/layers/util.py:82-100, 116-142: A.csr(), torch.sort(A.row), dglsp.from_csr(...).csc()), which the
 * reference counts inside a training epoch (DFGNN/script/train/train_batch_graph_timing.py:115-143).
 *   src, dst    node ids of the nnz edges (row = src, column = dst, DFGNN/layers/util.py:53-56); int64 when
 *               idx64 != 0 (DGL's default idtype), else int32; ids are clamped to [0, m)
 *   row_ptr int32[m+1], col_ind int32[nnz], rows int32[nnz]: CSR = stable sort by row (COO order kept inside a
 *               row), rows = the sorted row ids
 *   edge_order  int32[nnz]: COO position of each CSR slot (A.csr()'s value indices: val = A.val[edge_order])
 *   col_ptr int32[m+1], row_ind int32[nnz], val_idx int32[nnz]: CSC = stable sort of the CSR list by column,
 *               val_idx = CSR slot of each CSC entry; pass all three as NULL to skip the CSC half
 *   ws          device workspace of dfgnn_preprocess_ws_bytes(m, nnz) bytes
 * No host synchronisation, no allocation.  Two rocPRIM radix sorts over the ceil(log2 m) low key bits + three
 * small kernels. */
size_t dfgnn_preprocess_ws_bytes(int m, int nnz);
int dfgnn_preprocess_hyper(int m, int nnz, const void *src, const void *dst, int idx64, int *row_ptr,
                           int *col_ind, int *rows, int *edge_order, int *col_ptr, int *row_ind,
                           int *val_idx, void *ws, size_t ws_bytes, dfgnn_stream_t stream);

/* ---- GT (graph transformer) ------------------------------------------------------------------
 * replaces gt_hyper_inference  (DFGNN/src/fused_gtconv/fused_gtconv.cpp:278-314,
 *                               fused_gtconv_hyper.cu:679-725) when attn_edge == NULL, and
 *          gt_hyper_forward    (fused_gtconv.cpp:79-116, fused_gtconv_hyper.cu:727-760)
 *          when attn_edge != NULL (training forward: also writes the normalised attention).
 * plan / plan_meta: device plan + its 12 host header words from dfgnn_plan_build, or NULL/NULL.
 * edge_ws: scratch fp32[h, nnz], needed only for inference (attn_edge == NULL) with a plan whose
 *          meta[8] > 0 (ranges whose per-edge values do not fit LDS next to their rows); else NULL. */
int dfgnn_gt_hyper_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                       const int *rows, const float *val, const float *Q, const float *K,
                       const float *V, float *attn_edge, float *edge_ws, float *out,
                       const int *plan, const int *plan_meta, dfgnn_stream_t stream);

/* replaces gt_backward (fused_gtconv.cpp:125-172, fused_gtconv_backward.cu:193-265).
 * col_ptr int32[m+1], row_ind int32[nnz], val_idx int32[nnz] (CSR slot of each CSC entry).
 * grad_edge is caller-provided scratch fp32[h, nnz] (the reference allocates it inside); with a
 * plan, ranges that run LDS-resident keep dS on chip and leave their part of grad_edge untouched.
 * dQ, dK, dV are fully written (no pre-zeroing needed).  plan / plan_meta as in dfgnn_gt_hyper_fwd. */
int dfgnn_gt_bwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                 const int *rows, const float *val, const int *col_ptr, const int *row_ind,
                 const int *val_idx, const float *Q, const float *K, const float *V,
                 const float *attn_edge, const float *grad_out, float *grad_edge, float *dQ,
                 float *dK, float *dV, const int *plan, const int *plan_meta,
                 dfgnn_stream_t stream);

/* The statistics-saving form of the training pair above: what FusedGTFunction_hyper (DFGNN/operators/fused_gtconv.py:
 * 79-158) needs from its forward is enough to rebuild the attention in the backward, not the attention itself.  Instead
 * of attn_edge[h, nnz] (written by gt_hyper_forward, fused_gtconv_hyper.cu:146-149, read back by gt_backward,
 * fused_gtconv_backward.cu:132-136: 8 h nnz bytes through HBM) the forward saves, per (row, head), the logit maximum and
 * the sum of exponentials -- row_max, row_sum: fp32[m, h]; an empty row has row_max = -1e38, row_sum = 0 -- and the
 * backward recomputes P_e = exp(s_e - row_max) / row_sum with one more Q K^T product on the matrix cores.
 * Only for batches that the matrix-core kernels cover completely: dfgnn_gt_stats_applies(m, nnz, h, f, plan_meta) == 1
 * (host-only; every range of the plan dense, nothing spilled); otherwise both calls return DFGNN_E_UNSUPPORTED and the
 * caller uses dfgnn_gt_hyper_fwd / dfgnn_gt_bwd.  The sparse structure reaches these kernels through the plan's edge
 * bitmaps alone: rows, CSC arrays, grad_edge are not needed.
 * Edge values (the reference's `attn * val`, fused_gtconv_hyper.cu:88-90): `weights` = NULL for unit values, else the
 * values in the dense form of dfgnn_plan_dense_weights below (built once per (plan, val), e.g. per batch of a dataset
 * whose edge weights do not change): s_e = val_e <Q_i, K_j> in the forward, d s_e / d <Q_i, K_j> = val_e in the backward.
 * row_max = row_sum = NULL in the forward: nothing is saved (inference with edge values on the matrix cores).
 * Results equal the attn_edge form's (same arithmetic for out; dQ, dK, dV to fp32 rounding). */
int dfgnn_gt_stats_applies(int m, int nnz, int h, int f, const int *plan_meta);
int dfgnn_gt_hyper_fwd_stats(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *weights,
                             const float *Q, const float *K, const float *V, float *row_max, float *row_sum, float *out,
                             const int *plan, const int *plan_meta, dfgnn_stream_t stream);
int dfgnn_gt_bwd_stats(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *weights,
                       const float *Q, const float *K, const float *V, const float *row_max, const float *row_sum,
                       const float *grad_out, float *dQ, float *dK, float *dV, const int *plan, const int *plan_meta,
                       dfgnn_stream_t stream);
/* The attn_edge pair with the attention values in RANK order (one head, unit edge values, dfgnn_gt_stats_applies == 1).
 * attn_edge[h, nnz] between gt_hyper_forward and gt_backward is internal to FusedGTFunction_hyper
 * (DFGNN/operators/fused_gtconv.py:79-158): all the pair needs is that both sides agree on an order.  Here the value of
 * row i's k-th edge BY INCREASING COLUMN is stored at attn_ranked[row_ptr[i] + k] (fp32[nnz]): the forward then finds an
 * edge's slot from the plan's bitmap (the number of set bits before it) instead of loading the edge list and building a
 * position map, and the backward is dfgnn_gt_bwd's matrix-core kernel reading the plan's rank-ordered coordinates.  For
 * rows whose columns are already increasing attn_ranked == attn_edge.  Results equal dfgnn_gt_hyper_fwd / dfgnn_gt_bwd
 * (same arithmetic).  Other shapes: DFGNN_E_UNSUPPORTED. */
int dfgnn_gt_hyper_fwd_ranked(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *Q,
                              const float *K, const float *V, float *attn_ranked, float *out, const int *plan,
                              const int *plan_meta, dfgnn_stream_t stream);
int dfgnn_gt_bwd_ranked(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *Q,
                        const float *K, const float *V, const float *attn_ranked, const float *grad_out, float *dQ,
                        float *dK, float *dV, const int *plan, const int *plan_meta, dfgnn_stream_t stream);

/* weights[256 i + c] = val[e] for the edge e from node i to the c-th node of i's range of the plan, 0 elsewhere:
 * dfgnn_plan_dense_weights_floats(m) = 256 m floats (device, 16-byte aligned), written by one memset + one kernel on
 * `stream`.  val: fp32[nnz] in CSR order.  Only the dense ranges of the plan are filled (dfgnn_gt_stats_applies == 1:
 * all of them). */
size_t dfgnn_plan_dense_weights_floats(int m);
int dfgnn_plan_dense_weights(int m, int nnz, const int *row_ptr, const float *val, const int *plan, const int *plan_meta,
                             float *weights, dfgnn_stream_t stream);

/* The two launches of the plan-less dfgnn_gt_bwd, exposed separately so each can be timed / profiled on its own
 * (dfgnn_gt_bwd == rows pass then cols pass on the same stream):
 *   rows pass (CSR): dP = <dO[i],V[j]>, dS = P (dP - sum_row P dP) -> grad_edge, dQ   (fused_backward_kernel,
 *                    fused_gtconv_backward.cu:73-191)
 *   cols pass (CSC): dV[j] = sum P dO[i], dK[j] = sum dS val Q[i]                      (spmm_backward_kernel,
 *                    fused_gtconv_backward.cu:40-70) */
int dfgnn_gt_bwd_rows(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                      const int *rows, const float *val, const float *K, const float *V,
                      const float *attn_edge, const float *grad_out, float *grad_edge, float *dQ,
                      dfgnn_stream_t stream);
int dfgnn_gt_bwd_cols(int m, int nnz, int h, int f, const float *val, const int *col_ptr,
                      const int *row_ind, const int *val_idx, const float *Q, const float *attn_edge,
                      const float *grad_edge, const float *grad_out, float *dK, float *dV,
                      dfgnn_stream_t stream);

/* replaces gt_tiling_inference (fused_gtconv.cpp:244-276, fused_gtconv_tiling.cu:92-118):
 * CSR only, online softmax over fixed-size neighbour tiles, no degree limit. */
int dfgnn_gt_tiling_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                        const float *val, const float *Q, const float *K, const float *V,
                        float *out, dfgnn_stream_t stream);

/* replaces gt_softmax_inference (fused_gtconv.cpp:316-352, fused_gtconv_softmax.cu:10-54) and
 * gt_softmax_gm_inference (fused_gtconv.cpp:354-389, fused_gtconv_softmax_gm.cu:81-125):
 * two kernels, edge-parallel SDDMM into `logits` (caller scratch fp32[h, nnz]) then
 * node-parallel softmax+SpMM; the _gm form re-reads logits from global memory on every pass. */
int dfgnn_gt_softmax_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                         const int *rows, const float *val, const float *Q, const float *K,
                         const float *V, float *logits, float *out, dfgnn_stream_t stream);
int dfgnn_gt_softmax_gm_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                            const int *rows, const float *val, const float *Q, const float *K,
                            const float *V, float *logits, float *out, dfgnn_stream_t stream);

/* ---- GAT ---------------------------------------------------------------------------------------
 * attn_row, attn_col fp32[m, h]; X (in_feat) fp32[m, h, f].
 * replaces gat_inference_hyper (DFGNN/src/fused_gatconv/fused_gatconv.cpp:99-119,
 *                               fused_gatconv_hyper.cu:251-272)
 * plan / plan_meta / edge_ws as in dfgnn_gt_hyper_fwd (edge_ws is needed whenever meta[8] > 0). */
int dfgnn_gat_hyper_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                        const int *rows, const float *attn_row, const float *attn_col,
                        float negative_slope, const float *X, float *edge_ws, float *out,
                        const int *plan, const int *plan_meta, dfgnn_stream_t stream);

/* replaces gt_csr_inference / gt_csr_gm_inference (fused_gtconv.cpp:174-242; fused_gt_csr,
 * fused_gt_csr_global_memory, fused_gtconv_csr.cu:10-219): the node-parallel CSR baselines of the reference's sweeps --
 * a wave per row, the row's logits materialised (in LDS: 'csr', rows longer than the 2048-float buffer use `logits`;
 * in global memory: 'csr_gm'), then max / sum / weighted-sum sweeps.  logits: fp32[h, nnz] scratch. */
int dfgnn_gt_csr_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *val,
                     const float *Q, const float *K, const float *V, float *logits, float *out,
                     dfgnn_stream_t stream);
int dfgnn_gt_csr_gm_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind, const float *val,
                        const float *Q, const float *K, const float *V, float *logits, float *out,
                        dfgnn_stream_t stream);

/* replaces gat_inference_hyper_recompute (fused_gatconv.cpp:124-142; fused_gat_hyper_recompute_inference_vec4,
 * fused_gatconv_hyper_recompute.cu:118-216): node-parallel, no logit storage -- the rank-one logits are recomputed in
 * each of the three sweeps.  Any f (the reference exit(0)s unless f % 128 == 0). */
int dfgnn_gat_recompute_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                            const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                            float *out, dfgnn_stream_t stream);

/* replaces gat_inference_softmax (fused_gatconv.cpp:40-61, fused_gatconv_softmax.cu:33-56) and
 * gat_inference_softmax_gm (fused_gatconv.cpp:69-90, fused_gatconv_softmax_gm.cu) */
int dfgnn_gat_softmax_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                          const int *rows, const float *attn_row, const float *attn_col,
                          float negative_slope, const float *X, float *logits, float *out,
                          dfgnn_stream_t stream);
int dfgnn_gat_softmax_gm_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                             const int *rows, const float *attn_row, const float *attn_col,
                             float negative_slope, const float *X, float *logits, float *out,
                             dfgnn_stream_t stream);

/* replaces gat_inference_tiling (fused_gatconv.cpp:196-219, fused_gatconv_tiling.cu:78-103) */
int dfgnn_gat_tiling_fwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                         const float *attn_row, const float *attn_col, float negative_slope,
                         const float *X, float *out, dfgnn_stream_t stream);

/* dfgnn_gat_tiling_fwd for super-node full graphs whose feature table [m, h, f] does not fit the L2s (reddit: 119 MB): the
 * columns are cut into chunks of `chunk_rows` nodes (<= 32768; 8192 x 512 B = one XCD's 4 MiB L2), the edges are
 * re-ordered chunk-major, every (row, chunk) segment gives an online-softmax partial state and a second kernel merges
 * a row's states (csrc/gat_tiling_chunked.hip).  Same result as dfgnn_gat_tiling_fwd to fp32 rounding.
 *   seg_ptr  int32[nchunks * m + 1], nchunks = ceil(m / chunk_rows): the edges of row r into chunk c are
 *            seg_ptr[c m + r] .. seg_ptr[c m + r + 1] of the chunk-major edge order (= the CSR order sorted stably by
 *            the chunk of the column)
 *   ccol     int16[nnz]: column - chunk * chunk_rows of every edge in that order
 *   ws       device workspace of dfgnn_gat_tiling_chunked_ws_bytes(m, h, f, chunk_rows) bytes (partial states)
 * seg_ptr / ccol depend on the graph only: preprocessing, built once (the Python binding builds and caches them). */
size_t dfgnn_gat_tiling_chunked_ws_bytes(int m, int h, int f, int chunk_rows);
int dfgnn_gat_tiling_chunked_fwd(int m, int nnz, int h, int f, int chunk_rows, const int *seg_ptr, const short *ccol,
                                 const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                                 float *out, void *ws, size_t ws_bytes, dfgnn_stream_t stream);

/* First kernel of the reference's 'hyper_v2' variant (gat_inference_hyper_v2, fused_gatconv.cpp:148-158;
 * fused_gat_dot_attn_weight, fused_gatconv_hyper_v2.cu:212-249): the per-node attention scores from the layer's
 * attention vectors a_l, a_r fp32[h, f]:  attn_row[i, hd] = <a_l[hd], X[i, hd]>,  attn_col[i, hd] = <a_r[hd], X[i, hd]>.
 * X is read once for both.  The second kernel of 'hyper_v2' is dfgnn_gat_hyper_fwd / dfgnn_gat_tiling_fwd. */
int dfgnn_gat_attn_scores(int m, int h, int f, const float *a_l, const float *a_r, const float *X,
                          float *attn_row, float *attn_col, dfgnn_stream_t stream);

/* ---- GAT training pair (FusedGATFunction, DFGNN/operators/fused_gatconv.py:95-176) --------------------
 * edge_max, edge_sum fp32[m, h]: per-row maximum of the LeakyReLU logits (-1e38 for an empty row) and
 *   sum_e exp(s_e - max); the backward recomputes P_e from them, as the reference does.
 * edge_mask fp32[nnz, h], EDGE-major (index e*h + head, fused_gatconv_kernel.cu:101): uniform randoms for
 *   attention dropout, an INPUT here (the reference draws it inside with cuRAND seeded by clock(),
 *   fused_gatconv_kernel.cu:1074-1083; the binding draws it with torch.rand so runs are reproducible).
 *   Edge e of head hd is kept iff edge_mask[e*h + hd] > attn_drop and its attention is scaled by
 *   1 / (1 - attn_drop).  NULL = no dropout (attn_drop is then ignored); with a mask, 0 <= attn_drop < 1.
 * rows / plan / plan_meta: optional (NULL = general CSR kernels).  With all three, the plan's dense ranges
 *   (meta[9] of them: small dense member graphs of a batch) run on the matrix-core kernels, one workgroup per
 *   range as in dfgnn_gt_hyper_fwd / dfgnn_gt_bwd, and the general kernels cover only the remaining ranges.
 *   The reference's gat_forward / gat_backward take no COO rows: the binding derives them from row_ptr once per
 *   batch structure, next to the plan.
 *
 * dfgnn_gat_fwd_train replaces gat_forward (fused_gatconv.cpp:11-32, fused_gatconv_kernel.cu:24-125, 1062-1129):
 *   writes out[m, h, f], edge_max, edge_sum. */
int dfgnn_gat_fwd_train(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                        const int *rows, const float *attn_row, const float *attn_col,
                        float negative_slope, const float *X, const float *edge_mask, float attn_drop,
                        float *edge_max, float *edge_sum, float *out, const int *plan,
                        const int *plan_meta, dfgnn_stream_t stream);

/* replaces gat_backward (fused_gatconv.cpp:291-353, fused_gatconv_kernel.cu:609-865, 1172-1244).
 * col_ptr int32[m+1], row_ind int32[nnz], permute int32[nnz] (CSR slot of each CSC entry; the GT path calls
 * it val_idx).  grad_edge: caller scratch fp32[h, nnz] (untouched on the matrix-core path).  Writes grad_feat
 * fp32[m, h, f], grad_attn_row and grad_attn_col fp32[m, h] in full (no pre-zeroing, no atomics: the column
 * sums are deterministic). */
int dfgnn_gat_bwd(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                  const int *rows, const int *col_ptr, const int *row_ind, const int *permute,
                  const float *attn_row, const float *attn_col, float negative_slope, const float *X,
                  const float *edge_max, const float *edge_sum, const float *edge_mask,
                  float attn_drop, const float *grad_out, float *grad_edge, float *grad_feat,
                  float *grad_attn_row, float *grad_attn_col, const int *plan, const int *plan_meta,
                  dfgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DFGNN_H_ */
