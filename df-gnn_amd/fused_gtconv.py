"""`fused_gtconv` -- drop-in for the reference's extension module of the same name
(PYBIND11_MODULE(fused_gtconv), DFGNN/src/fused_gtconv/fused_gtconv.cpp:577-602), bound to the
MI355X HIP library through its C ABI (include/dfgnn.h).

Same function names, positional signatures, return conventions (list vs bare tensor) and error
type (RuntimeError) as the reference.  Differences, all deliberate:
  * kernels launch on torch's *current* stream of the tensors' device (reference: legacy default
    stream, no device guard);
  * dtype / shape checks are real (reference: asserts compiled out);
  * `smem_consume` is accepted and ignored: LDS sizing is decided per workgroup inside the
    kernels, with an online-softmax fallback instead of the reference's silent overflow.
"""
import torch

import dfgnn_native as _n
import os

from _binding_util import (as_int32, check_contiguous, check_device, check_dtype, check_feat3, get_plan, get_plan_obj,
                           plan_dense_weights, ptr,
                           stream_ptr, val_ptr)

# Set to False to force the general (plan-less) kernels; results are identical either way.
USE_BLOCK_PLAN = True


def _dims(indptr, indices, Q):
    return indptr.size(0) - 1, indices.size(0), Q.size(1), Q.size(2)


def _check_graph(indptr, indices, m_feat):
    if indptr.dim() != 1 or indices.dim() != 1:
        raise RuntimeError("indptr / indices must be 1-D")
    if indptr.size(0) - 1 != m_feat:
        raise RuntimeError(f"indptr describes {indptr.size(0) - 1} rows but features have {m_feat} nodes")


def _check_qkv(Q, K, V):
    check_device(Q=Q, K=K, V=V)
    check_contiguous(Q=Q, K=K, V=V)
    check_dtype(torch.float32, Q=Q, K=K, V=V)
    check_feat3(Q=Q, K=K, V=V)


def _check_edges(nnz, **arrs):
    for name, t in arrs.items():
        if t.dim() != 1 or t.size(0) != nnz:
            raise RuntimeError(f"{name} must have shape ({nnz},), got {tuple(t.shape)}")


def gt_hyper_inference(indptr, indices, rows, val, smem_consume, Q, K, V):
    """fused_gtconv.cpp:278-314 -> [out]"""
    if val_ptr(val) is not None and Q.dim() == 3 and Q.is_cuda:
        # edge values: the matrix-core forward reads them in the plan's dense form (csrc/gt_dense_stats_w.hip) when
        # every range of the batch is dense; anything else takes the edge-walking kernels below
        plan = gt_stats_pair_applies(indptr, indices, val, Q)
        if plan is not None:
            return gt_hyper_forward_stats(indptr, indices, Q, K, V, plan=plan, val=val, save_stats=False)
    ext = _n.ext()
    if ext is not None:  # torch C++ binding (csrc/torch_ext.cpp): same checks, same C ABI call
        plan, meta, _ = get_plan(indptr, indices, Q.size(-1) if Q.dim() == 3 else 0, USE_BLOCK_PLAN)
        return ext.gt_hyper_fwd(indptr, indices, rows, val, Q, K, V, False, val_ptr(val) is None, plan or 0, meta or 0)
    check_device(indptr=indptr, indices=indices, rows=rows, val=val)
    check_contiguous(indptr=indptr, indices=indices, rows=rows, val=val)
    check_dtype(torch.int32, indptr=indptr, indices=indices, rows=rows)
    check_dtype(torch.float32, val=val)
    _check_qkv(Q, K, V)
    m, nnz, h, f = _dims(indptr, indices, Q)
    _check_graph(indptr, indices, Q.size(0))
    _check_edges(nnz, rows=rows, val=val)
    with torch.cuda.device(Q.device):
        out = torch.empty_like(Q)
        plan, meta, need_ws = get_plan(indptr, indices, f, USE_BLOCK_PLAN)
        # scratch for per-edge values: required when the plan has edge-global ranges
        ws = torch.empty((h, nnz), dtype=torch.float32, device=Q.device) if plan is not None else None
        _n.check(_n.lib().dfgnn_gt_hyper_fwd(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(rows), val_ptr(val),
                                             ptr(Q), ptr(K), ptr(V), None, ptr(ws), ptr(out), plan, meta,
                                             stream_ptr(Q.device)), "gt_hyper_inference")
    return [out]


def gt_hyper_inference_ablation(indptr, indices, rows, val, smem_consume, Q, K, V):
    """fused_gtconv.cpp:538-575.  The reference's de-optimised ablation kernels are a paper study
    (out of scope); the entry point is kept and runs the production kernel."""
    return gt_hyper_inference(indptr, indices, rows, val, smem_consume, Q, K, V)


def gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem_consume, Q, K, V):
    """fused_gtconv.cpp:79-116 -> [out, attn_edge[h, nnz]] (training forward)."""
    ext = _n.ext()
    if ext is not None:
        plan, meta, _ = get_plan(row_ptr, col_ind, Q.size(-1) if Q.dim() == 3 else 0, USE_BLOCK_PLAN)
        return ext.gt_hyper_fwd(row_ptr, col_ind, rows, val, Q, K, V, True, val_ptr(val) is None, plan or 0, meta or 0)
    check_device(row_ptr=row_ptr, col_ind=col_ind, val=val, rows=rows)
    check_contiguous(row_ptr=row_ptr, col_ind=col_ind, val=val, rows=rows)
    check_dtype(torch.int32, row_ptr=row_ptr, col_ind=col_ind, rows=rows)
    check_dtype(torch.float32, val=val)
    _check_qkv(Q, K, V)
    m, nnz, h, f = _dims(row_ptr, col_ind, Q)
    _check_graph(row_ptr, col_ind, Q.size(0))
    _check_edges(nnz, rows=rows, val=val)
    with torch.cuda.device(Q.device):
        out = torch.empty_like(Q)
        attn_edge = torch.empty((h, nnz), dtype=torch.float32, device=Q.device)
        plan, meta, _ = get_plan(row_ptr, col_ind, f, USE_BLOCK_PLAN)
        _n.check(_n.lib().dfgnn_gt_hyper_fwd(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(rows), val_ptr(val),
                                             ptr(Q), ptr(K), ptr(V), ptr(attn_edge), None, ptr(out), plan, meta,
                                             stream_ptr(Q.device)), "gt_hyper_forward")
    return [out, attn_edge]


def gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem_consume, Q, K, V, attn_edge,
                grad):
    """fused_gtconv.cpp:125-172 -> [dQ, dK, dV]."""
    val_idx = as_int32(val_idx)
    ext = _n.ext()
    if ext is not None:
        plan, meta, _ = get_plan(row_ptr, col_ind, Q.size(-1) if Q.dim() == 3 else 0, USE_BLOCK_PLAN)
        return ext.gt_bwd(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, Q, K, V, attn_edge, grad,
                          val_ptr(val) is None, plan or 0, meta or 0)
    check_device(row_ptr=row_ptr, col_ind=col_ind, rows=rows, val=val, col_ptr=col_ptr, row_ind=row_ind,
                 val_idx=val_idx, attn_edge=attn_edge, grad=grad)
    check_contiguous(row_ptr=row_ptr, col_ind=col_ind, rows=rows, val=val, col_ptr=col_ptr, row_ind=row_ind,
                     val_idx=val_idx, attn_edge=attn_edge, grad=grad)
    check_dtype(torch.int32, row_ptr=row_ptr, col_ind=col_ind, rows=rows, col_ptr=col_ptr, row_ind=row_ind)
    check_dtype(torch.float32, val=val, attn_edge=attn_edge, grad=grad)
    _check_qkv(Q, K, V)
    check_feat3(Q=Q, grad=grad)
    m, nnz, h, f = _dims(row_ptr, col_ind, Q)
    _check_graph(row_ptr, col_ind, Q.size(0))
    _check_edges(nnz, rows=rows, val=val, row_ind=row_ind, val_idx=val_idx)
    if col_ptr.dim() != 1 or col_ptr.size(0) != m + 1:
        raise RuntimeError(f"col_ptr must have shape ({m + 1},): the adjacency must be square")
    if attn_edge.numel() != h * nnz:
        raise RuntimeError(f"attn_edge must have {h}*{nnz} elements, got {attn_edge.numel()}")
    with torch.cuda.device(Q.device):
        grad_edge = torch.empty((h, nnz), dtype=torch.float32, device=Q.device)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        plan, meta, _ = get_plan(row_ptr, col_ind, f, USE_BLOCK_PLAN)
        _n.check(_n.lib().dfgnn_gt_bwd(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(rows), val_ptr(val),
                                       ptr(col_ptr), ptr(row_ind), ptr(val_idx), ptr(Q), ptr(K), ptr(V),
                                       ptr(attn_edge), ptr(grad), ptr(grad_edge), ptr(dQ), ptr(dK), ptr(dV),
                                       plan, meta, stream_ptr(Q.device)), "gt_backward")
    return [dQ, dK, dV]


# ---- the statistics-saving training pair (include/dfgnn.h: dfgnn_gt_hyper_fwd_stats / dfgnn_gt_bwd_stats) ------------
# Not part of the reference's module: FusedGTFunction_hyper (DFGNN/operators/fused_gtconv.py) takes this pair instead of
# gt_hyper_forward / gt_backward when the whole batch runs on the matrix-core kernels -- same results, no attn_edge.
# Which pair the autograd Function takes is a measured choice: the statistics pair saves the attn_edge
# round trip (8 h nnz bytes) and pays one more K image pass in the backward -- from two heads on it is the faster one
# (8 x 16 heads: 0.58 against 0.65 ms per step), at one head the attn_edge pair is (0.263 against 0.267 ms).
# DFGNN_STATS in the environment: 0 = the attn_edge pair everywhere, 1 = the statistics pair wherever it applies (A/B runs).
_STATS_ENV = os.environ.get("DFGNN_STATS", "auto")
USE_STATS_PAIR = _STATS_ENV != "0"
STATS_PAIR_MIN_HEADS = 1 if _STATS_ENV == "1" else 2
# the attn_edge pair in rank order (further down): DFGNN_RANKED=0 switches it off, DFGNN_RANKED_HEADS lists the head counts
# it is taken at (default: one head)
USE_RANKED_PAIR = os.environ.get("DFGNN_RANKED", "1") != "0"
RANKED_HEADS = tuple(int(x) for x in os.environ.get("DFGNN_RANKED_HEADS", "1").split(",") if x)


def gt_stats_pair_applies(row_ptr, col_ind, val, Q):
    """The block plan (a true value) when gt_hyper_forward_stats / gt_backward_stats can serve this call -- a plan whose
    ranges are all dense (dfgnn_gt_stats_applies) -- else None.  The plan may be handed to the two calls (`plan=`), which
    then skip their own look-up; edge values other than ones go along as `val=` (the calls turn them into the plan's
    dense weights, _binding_util.plan_dense_weights)."""
    if not (USE_STATS_PAIR and USE_BLOCK_PLAN) or Q.dim() != 3 or not Q.is_cuda:
        return None
    plan = get_plan_obj(row_ptr, col_ind, Q.size(-1), True)
    if plan is not None and plan.stats_applies(Q.size(1)):
        return plan
    return None


def gt_stats_pair_chosen(row_ptr, col_ind, val, Q):
    """gt_stats_pair_applies under the policy above: the plan when the autograd Function should take the statistics pair
    for this call, else None.  With edge values it is always the pair to take where it applies: the attn_edge pair has
    no matrix-core form for them."""
    if Q.dim() != 3 or (Q.size(1) < STATS_PAIR_MIN_HEADS and val_ptr(val) is None):
        return None
    if _STATS_ENV != "1" and Q.size(1) in RANKED_HEADS and gt_ranked_pair_chosen(row_ptr, col_ind, val, Q) is not None:
        return None     # (a head count handed to the rank-ordered attn_edge pair: DFGNN_RANKED_HEADS)
    return gt_stats_pair_applies(row_ptr, col_ind, val, Q)


def _stats_weights(plan, row_ptr, val):
    """None for unit edge values (or no `val`), else the plan's dense weights of `val`."""
    if val is None or val_ptr(val) is None:
        return None
    return plan_dense_weights(plan, row_ptr, val)


# ---- the attn_edge pair in rank order (include/dfgnn.h: dfgnn_gt_hyper_fwd_ranked / dfgnn_gt_bwd_ranked) --------------
# What FusedGTFunction_hyper takes at ONE head on an all-dense batch with unit edge values: the attention values travel from
# forward to backward like in the reference, but ordered by column within a row, which lets the forward find an edge's slot
# from the plan's bitmap instead of the edge list (csrc/gt_dense_stats_w.hip: gt_dense_fwd_ranked_kernel).


def gt_ranked_pair_applies(row_ptr, col_ind, val, Q):
    """The block plan when gt_hyper_forward_ranked / gt_backward_ranked can serve this call, else None."""
    if not (USE_RANKED_PAIR and USE_BLOCK_PLAN) or Q.dim() != 3 or not Q.is_cuda or val_ptr(val) is not None:
        return None
    plan = get_plan_obj(row_ptr, col_ind, Q.size(-1), True)
    if plan is not None and plan.stats_applies(Q.size(1)):
        return plan
    return None


def gt_ranked_pair_chosen(row_ptr, col_ind, val, Q):
    """gt_ranked_pair_applies under the measured policy: at one head (C3: forward 95 -> 85 us; batches without a range of
    more than 128 nodes take the 256-thread form of the same forward, two workgroups per CU).  With several heads the
    rank-ordered forward times like the CSR-ordered one and the statistics pair is the faster choice; RANKED_HEADS widens
    the head counts for A/B runs (DFGNN_RANKED_HEADS=1,2,4)."""
    if Q.dim() != 3 or Q.size(1) not in RANKED_HEADS:
        return None
    return gt_ranked_pair_applies(row_ptr, col_ind, val, Q)


def gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V, plan=None):
    """-> [out, attn_ranked[1, nnz]] (row i's k-th edge by increasing column at row_ptr[i] + k)."""
    if plan is None:
        plan = get_plan_obj(row_ptr, col_ind, Q.size(-1) if Q.dim() == 3 else 0, USE_BLOCK_PLAN)
    pp, mp = plan.ptrs() if plan is not None else (None, None)
    ext = _n.ext()
    if ext is not None and hasattr(ext, "gt_hyper_fwd_ranked"):
        return ext.gt_hyper_fwd_ranked(row_ptr, col_ind, Q, K, V, pp or 0, mp or 0)
    check_device(row_ptr=row_ptr, col_ind=col_ind)
    check_contiguous(row_ptr=row_ptr, col_ind=col_ind)
    check_dtype(torch.int32, row_ptr=row_ptr, col_ind=col_ind)
    _check_qkv(Q, K, V)
    m, nnz, h, f = _dims(row_ptr, col_ind, Q)
    _check_graph(row_ptr, col_ind, Q.size(0))
    with torch.cuda.device(Q.device):
        out = torch.empty_like(Q)
        attn = torch.empty((h, nnz), dtype=torch.float32, device=Q.device)
        _n.check(_n.lib().dfgnn_gt_hyper_fwd_ranked(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(Q), ptr(K), ptr(V), ptr(attn),
                                                    ptr(out), pp, mp, stream_ptr(Q.device)), "gt_hyper_forward_ranked")
    return [out, attn]


def gt_backward_ranked(row_ptr, col_ind, Q, K, V, attn_ranked, grad, plan=None):
    """-> [dQ, dK, dV] from the rank-ordered attention values of gt_hyper_forward_ranked."""
    if plan is None:
        plan = get_plan_obj(row_ptr, col_ind, Q.size(-1) if Q.dim() == 3 else 0, USE_BLOCK_PLAN)
    pp, mp = plan.ptrs() if plan is not None else (None, None)
    ext = _n.ext()
    if ext is not None and hasattr(ext, "gt_bwd_ranked"):
        return ext.gt_bwd_ranked(row_ptr, col_ind, Q, K, V, attn_ranked, grad, pp or 0, mp or 0)
    check_device(row_ptr=row_ptr, col_ind=col_ind, attn_ranked=attn_ranked, grad=grad)
    check_contiguous(row_ptr=row_ptr, col_ind=col_ind, attn_ranked=attn_ranked, grad=grad)
    check_dtype(torch.int32, row_ptr=row_ptr, col_ind=col_ind)
    check_dtype(torch.float32, attn_ranked=attn_ranked, grad=grad)
    _check_qkv(Q, K, V)
    check_feat3(Q=Q, grad=grad)
    m, nnz, h, f = _dims(row_ptr, col_ind, Q)
    _check_graph(row_ptr, col_ind, Q.size(0))
    if attn_ranked.numel() != h * nnz:
        raise RuntimeError(f"attn_ranked must have {h}*{nnz} elements, got {attn_ranked.numel()}")
    with torch.cuda.device(Q.device):
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        _n.check(_n.lib().dfgnn_gt_bwd_ranked(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(Q), ptr(K), ptr(V), ptr(attn_ranked),
                                              ptr(grad), ptr(dQ), ptr(dK), ptr(dV), pp, mp, stream_ptr(Q.device)),
                 "gt_backward_ranked")
    return [dQ, dK, dV]


def gt_hyper_step_raw(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, grad):
    """-> [out, dQ, dK, dV]: the launches of one FusedGTFunction_hyper forward + backward as explicit operator calls (no
    autograd graph), choosing the pair the way the autograd function does -- what the HIP-graph replays capture."""
    plan = gt_stats_pair_chosen(row_ptr, col_ind, val, Q)
    if plan is not None:
        out, rmax, rsum = gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V, plan=plan, val=val)
        return [out] + list(gt_backward_stats(row_ptr, col_ind, Q, K, V, rmax, rsum, grad, plan=plan, val=val))
    plan = gt_ranked_pair_chosen(row_ptr, col_ind, val, Q)
    if plan is not None:
        out, attn = gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V, plan=plan)
        return [out] + list(gt_backward_ranked(row_ptr, col_ind, Q, K, V, attn, grad, plan=plan))
    out, attn = gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    return [out] + list(gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, grad))


def gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V, plan=None, val=None, save_stats=True):
    """-> [out, row_max[m, h], row_sum[m, h]]: the training forward without attn_edge (call gt_stats_pair_applies first).
    val: edge values (None or all ones: unit values).  save_stats=False -> [out] (inference)."""
    ext = _n.ext()
    if plan is None:
        plan = get_plan_obj(row_ptr, col_ind, Q.size(-1) if Q.dim() == 3 else 0, USE_BLOCK_PLAN)
    weights = _stats_weights(plan, row_ptr, val) if plan is not None else None
    plan, meta = plan.ptrs() if plan is not None else (None, None)
    if ext is not None and hasattr(ext, "gt_hyper_fwd_stats"):
        return ext.gt_hyper_fwd_stats(row_ptr, col_ind, Q, K, V, plan or 0, meta or 0, weights, save_stats)
    check_device(row_ptr=row_ptr, col_ind=col_ind)
    check_contiguous(row_ptr=row_ptr, col_ind=col_ind)
    check_dtype(torch.int32, row_ptr=row_ptr, col_ind=col_ind)
    _check_qkv(Q, K, V)
    m, nnz, h, f = _dims(row_ptr, col_ind, Q)
    _check_graph(row_ptr, col_ind, Q.size(0))
    with torch.cuda.device(Q.device):
        out = torch.empty_like(Q)
        row_max = torch.empty((m, h), dtype=torch.float32, device=Q.device) if save_stats else None
        row_sum = torch.empty((m, h), dtype=torch.float32, device=Q.device) if save_stats else None
        _n.check(_n.lib().dfgnn_gt_hyper_fwd_stats(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(weights), ptr(Q), ptr(K),
                                                   ptr(V), ptr(row_max), ptr(row_sum), ptr(out), plan, meta,
                                                   stream_ptr(Q.device)), "gt_hyper_forward_stats")
    return [out, row_max, row_sum] if save_stats else [out]


def gt_backward_stats(row_ptr, col_ind, Q, K, V, row_max, row_sum, grad, plan=None, val=None):
    """-> [dQ, dK, dV] from the row statistics of gt_hyper_forward_stats (P is recomputed on the matrix cores)."""
    ext = _n.ext()
    if plan is None:
        plan = get_plan_obj(row_ptr, col_ind, Q.size(-1) if Q.dim() == 3 else 0, USE_BLOCK_PLAN)
    weights = _stats_weights(plan, row_ptr, val) if plan is not None else None
    plan, meta = plan.ptrs() if plan is not None else (None, None)
    if ext is not None and hasattr(ext, "gt_bwd_stats"):
        return ext.gt_bwd_stats(row_ptr, col_ind, Q, K, V, row_max, row_sum, grad, plan or 0, meta or 0, weights)
    check_device(row_ptr=row_ptr, col_ind=col_ind, row_max=row_max, row_sum=row_sum, grad=grad)
    check_contiguous(row_ptr=row_ptr, col_ind=col_ind, row_max=row_max, row_sum=row_sum, grad=grad)
    check_dtype(torch.int32, row_ptr=row_ptr, col_ind=col_ind)
    check_dtype(torch.float32, row_max=row_max, row_sum=row_sum, grad=grad)
    _check_qkv(Q, K, V)
    check_feat3(Q=Q, grad=grad)
    m, nnz, h, f = _dims(row_ptr, col_ind, Q)
    _check_graph(row_ptr, col_ind, Q.size(0))
    for name, t in (("row_max", row_max), ("row_sum", row_sum)):
        if tuple(t.shape) != (m, h):
            raise RuntimeError(f"{name} must have shape ({m}, {h}), got {tuple(t.shape)}")
    with torch.cuda.device(Q.device):
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(K), torch.empty_like(V)
        _n.check(_n.lib().dfgnn_gt_bwd_stats(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(weights), ptr(Q), ptr(K), ptr(V),
                                             ptr(row_max), ptr(row_sum), ptr(grad), ptr(dQ), ptr(dK), ptr(dV), plan, meta,
                                             stream_ptr(Q.device)), "gt_backward_stats")
    return [dQ, dK, dV]


def _ext_variant(which, indptr, indices, rows, val, Q, K, V):
    """The CSR-taking inference variants through the torch C++ binding (csrc/torch_ext.cpp: gt_variant_fwd), or None."""
    ext = _n.ext()
    if ext is None or not hasattr(ext, "gt_variant_fwd"):
        return None
    return ext.gt_variant_fwd(which, indptr, indices, rows, val, Q, K, V, val_ptr(val) is None)


def gt_tiling_inference(indptr, indices, val, smem_consume, Q, K, V):
    """fused_gtconv.cpp:244-276 -> [out]"""
    out = _ext_variant(0, indptr, indices, None, val, Q, K, V)
    if out is not None:
        return [out]
    check_device(indptr=indptr, indices=indices, val=val)
    check_contiguous(indptr=indptr, indices=indices, val=val)
    check_dtype(torch.int32, indptr=indptr, indices=indices)
    check_dtype(torch.float32, val=val)
    _check_qkv(Q, K, V)
    m, nnz, h, f = _dims(indptr, indices, Q)
    _check_graph(indptr, indices, Q.size(0))
    _check_edges(nnz, val=val)
    with torch.cuda.device(Q.device):
        out = torch.empty_like(Q)
        _n.check(_n.lib().dfgnn_gt_tiling_fwd(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(val), ptr(Q), ptr(K),
                                              ptr(V), ptr(out), stream_ptr(Q.device)), "gt_tiling_inference")
    return [out]


def _gt_csr(fn_name, what, indptr, indices, val, Q, K, V):
    out = _ext_variant(1 if fn_name == "dfgnn_gt_csr_fwd" else 2, indptr, indices, None, val, Q, K, V)
    if out is not None:
        return [out]
    check_device(indptr=indptr, indices=indices, val=val)
    check_contiguous(indptr=indptr, indices=indices, val=val)
    check_dtype(torch.int32, indptr=indptr, indices=indices)
    check_dtype(torch.float32, val=val)
    _check_qkv(Q, K, V)
    m, nnz, h, f = _dims(indptr, indices, Q)
    _check_graph(indptr, indices, Q.size(0))
    _check_edges(nnz, val=val)
    with torch.cuda.device(Q.device):
        out = torch.empty_like(Q)
        logits = torch.empty((h, nnz), dtype=torch.float32, device=Q.device)
        _n.check(getattr(_n.lib(), fn_name)(m, nnz, h, f, ptr(indptr), ptr(indices), val_ptr(val), ptr(Q), ptr(K), ptr(V),
                                            ptr(logits), ptr(out), stream_ptr(Q.device)), what)
    return [out]


def gt_csr_inference(indptr, indices, val, smem_consume, Q, K, V):
    """fused_gtconv.cpp:174-207 -> [out].  The node-parallel CSR baseline of the reference's sweeps (fused_gt_csr): a wave
    per row, the row's logits materialised in LDS (csrc/csr_fwd.hip), then max / sum / weighted-sum sweeps."""
    return _gt_csr("dfgnn_gt_csr_fwd", "gt_csr_inference", indptr, indices, val, Q, K, V)


def gt_csr_gm_inference(indptr, indices, val, Q, K, V):
    """fused_gtconv.cpp:209-242 -> [out].  As gt_csr_inference with the logits in global memory
    (fused_gt_csr_global_memory)."""
    return _gt_csr("dfgnn_gt_csr_gm_fwd", "gt_csr_gm_inference", indptr, indices, val, Q, K, V)


def _gt_softmax(fn_name, what, indptr, indices, rows, val, Q, K, V):
    out = _ext_variant(3 if fn_name == "dfgnn_gt_softmax_fwd" else 4, indptr, indices, rows, val, Q, K, V)
    if out is not None:
        return out
    check_device(indptr=indptr, indices=indices, rows=rows, val=val)
    check_contiguous(indptr=indptr, indices=indices, rows=rows, val=val)
    check_dtype(torch.int32, indptr=indptr, indices=indices, rows=rows)
    check_dtype(torch.float32, val=val)
    _check_qkv(Q, K, V)
    m, nnz, h, f = _dims(indptr, indices, Q)
    _check_graph(indptr, indices, Q.size(0))
    _check_edges(nnz, rows=rows, val=val)
    with torch.cuda.device(Q.device):
        out = torch.empty_like(Q)
        logits = torch.empty((h, nnz), dtype=torch.float32, device=Q.device)
        _n.check(getattr(_n.lib(), fn_name)(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(rows), ptr(val), ptr(Q),
                                            ptr(K), ptr(V), ptr(logits), ptr(out), stream_ptr(Q.device)), what)
    return out


def gt_softmax_inference(indptr, indices, rows, val, smem_consume, Q, K, V):
    """fused_gtconv.cpp:316-352 -> [out] (two kernels: COO SDDMM, then LDS softmax+SpMM)."""
    return [_gt_softmax("dfgnn_gt_softmax_fwd", "gt_softmax_inference", indptr, indices, rows, val, Q, K, V)]


def gt_softmax_gm_inference(indptr, indices, rows, val, Q, K, V):
    """fused_gtconv.cpp:354-389 -> bare Tensor (two kernels, logits re-read from global memory)."""
    return _gt_softmax("dfgnn_gt_softmax_gm_fwd", "gt_softmax_gm_inference", indptr, indices, rows, val, Q, K, V)
