"""`fused_gatconv` -- drop-in for the reference's extension module of the same name
(PYBIND11_MODULE(fused_gatconv), DFGNN/src/fused_gatconv/fused_gatconv.cpp:355-372), bound to the
MI355X HIP library through its C ABI (include/dfgnn.h).  See fused_gtconv.py for the conventions.

In scope: the four inference entry points of the hyper / softmax /
softmax_gm / tiling variants.  The dgNN-inherited training pair gat_forward / gat_backward and the
experimental hyper_v2 / hyper_recompute / tb entry points are "next"  and raise
NotImplementedError rather than pretending.
"""
import torch

import dfgnn_native as _n
from _binding_util import check_contiguous, check_device, check_dtype, get_plan, ptr, stream_ptr

# Set to False to force the general (plan-less) kernels; results are identical either way.
USE_BLOCK_PLAN = True


def _check(attn_row, attn_col, indptr, indices, rows, in_feat):
    tensors = dict(attn_row=attn_row, attn_col=attn_col, indptr=indptr, indices=indices, in_feat=in_feat)
    if rows is not None:
        tensors["rows"] = rows
    check_device(**tensors)
    check_contiguous(**tensors)
    check_dtype(torch.float32, attn_row=attn_row, attn_col=attn_col, in_feat=in_feat)
    check_dtype(torch.int32, indptr=indptr, indices=indices)
    if rows is not None:
        check_dtype(torch.int32, rows=rows)
    if in_feat.dim() != 3:
        raise RuntimeError(f"in_feat must have shape [nodes, heads, feat], got {tuple(in_feat.shape)}")
    m, nnz = indptr.size(0) - 1, indices.size(0)
    h, f = attn_row.size(1) if attn_row.dim() == 2 else -1, in_feat.size(2)
    if tuple(attn_row.shape) != (m, in_feat.size(1)) or tuple(attn_col.shape) != (m, in_feat.size(1)):
        raise RuntimeError(f"attn_row / attn_col must have shape ({m}, {in_feat.size(1)}), got "
                           f"{tuple(attn_row.shape)} / {tuple(attn_col.shape)}")
    if in_feat.size(0) != m:
        raise RuntimeError(f"indptr describes {m} rows but in_feat has {in_feat.size(0)} nodes")
    if rows is not None and (rows.dim() != 1 or rows.size(0) != nnz):
        raise RuntimeError(f"rows must have shape ({nnz},), got {tuple(rows.shape)}")
    return m, nnz, h, f


def gat_inference_hyper(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """fused_gatconv.cpp:99-119 -> Tensor out[m, h, f]"""
    m, nnz, h, f = _check(attn_row, attn_col, indptr, indices, rows, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty_like(in_feat)
        plan, meta, need_ws = get_plan(indptr, indices, f, USE_BLOCK_PLAN)
        ws = torch.empty((h, nnz), dtype=torch.float32, device=in_feat.device) if need_ws else None
        _n.check(_n.lib().dfgnn_gat_hyper_fwd(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(rows), ptr(attn_row),
                                              ptr(attn_col), float(negative_slope), ptr(in_feat), ptr(ws), ptr(out),
                                              plan, meta, stream_ptr(in_feat.device)), "gat_inference_hyper")
    return out


def gat_inference_hyper_ablation(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope,
                                 in_feat):
    """fused_gatconv.cpp (ablation study entry): served by the production kernel."""
    return gat_inference_hyper(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat)


def _softmax(fn_name, what, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    m, nnz, h, f = _check(attn_row, attn_col, indptr, indices, rows, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty_like(in_feat)
        logits = torch.empty((h, nnz), dtype=torch.float32, device=in_feat.device)
        _n.check(getattr(_n.lib(), fn_name)(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(rows), ptr(attn_row),
                                            ptr(attn_col), float(negative_slope), ptr(in_feat), ptr(logits),
                                            ptr(out), stream_ptr(in_feat.device)), what)
    return out


def gat_inference_softmax(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """fused_gatconv.cpp:40-61 -> Tensor"""
    return _softmax("dfgnn_gat_softmax_fwd", "gat_inference_softmax", attn_row, attn_col, indptr, indices, rows,
                    negative_slope, in_feat)


def gat_inference_softmax_gm(attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """fused_gatconv.cpp:69-90 -> Tensor"""
    return _softmax("dfgnn_gat_softmax_gm_fwd", "gat_inference_softmax_gm", attn_row, attn_col, indptr, indices,
                    rows, negative_slope, in_feat)


def gat_inference_tiling(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """fused_gatconv.cpp:196-219 -> Tensor"""
    m, nnz, h, f = _check(attn_row, attn_col, row_ptr, col_ind, None, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty_like(in_feat)
        _n.check(_n.lib().dfgnn_gat_tiling_fwd(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(attn_row),
                                               ptr(attn_col), float(negative_slope), ptr(in_feat), ptr(out),
                                               stream_ptr(in_feat.device)), "gat_inference_tiling")
    return out


def gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """fused_gatconv.cpp (dgNN node-parallel CSR inference): same function as the
    tiling kernel, which serves it."""
    return gat_inference_tiling(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat)


def _next(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            f"fused_gatconv.{name} is outside this build's hot-path scope (SURVEY.md 8f 'next'); "
            "use the hyper / softmax / softmax_gm / tiling inference entry points")
    fn.__name__ = name
    return fn


gat_forward = _next("gat_forward")
gat_backward = _next("gat_backward")
gat_forward_tb = _next("gat_forward_tb")
gat_inference_hyper_v2 = _next("gat_inference_hyper_v2")
gat_inference_hyper_recompute = _next("gat_inference_hyper_recompute")
