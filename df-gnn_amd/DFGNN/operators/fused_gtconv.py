"""GT operator surface -- same names / signatures as the reference's DFGNN/operators/fused_gtconv.py.

Each function forwards to the `fused_gtconv` binding module (MI355X HIP kernels behind the C ABI).
Return conventions follow the reference: the binding returns a list for most entry points and the
wrapper unwraps element 0; `softmax_gm` returns a bare tensor (reference :279).
Note the argument-order quirk kept from the reference: `GTConvFuse_hyper` takes `rows` first, the
binding takes `row_ptr` first (reference :51-107).
"""
import fused_gtconv as fused_gt
import torch


def GTConvFuse_inference_hyper(indptr, indices, rows, val, smem_consume, Q, K, V):
    """hyper: one kernel, CSR + COO.  reference :5-25"""
    return fused_gt.gt_hyper_inference(indptr, indices, rows, val, smem_consume, Q, K, V)[0]


def GTConvFuse_inference_hyper_ablation(indptr, indices, rows, val, smem_consume, Q, K, V):
    """reference :28-48"""
    return fused_gt.gt_hyper_inference_ablation(indptr, indices, rows, val, smem_consume, Q, K, V)[0]


class FusedGTFunction_hyper(torch.autograd.Function):
    """Fused forward + fused backward.  reference :79-158

    The reference's forward saves the normalised attention (attn_edge[h, nnz]) for the backward.  When the whole batch
    runs on the matrix-core kernels (fused_gt.gt_stats_pair_applies: a block plan of dense ranges) and has at least two
    heads or edge values other than ones (fused_gt.gt_stats_pair_chosen: there it is the faster pair, measured) the
    forward saves two floats per (row, head) instead -- logit maximum and sum of exponentials -- and the backward
    recomputes the attention (include/dfgnn.h: dfgnn_gt_hyper_fwd_stats / dfgnn_gt_bwd_stats): same gradients, 8 h nnz
    bytes less through HBM.  At one head with unit values such a batch keeps the reference's form but with attn_edge in
    RANK order (by column within a row: fused_gt.gt_ranked_pair_applies), which spares the forward the edge list and the
    position map.  Any other batch takes the reference's form below them."""

    @staticmethod
    def forward(ctx, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume, Q, K, V):
        ctx.smem = smem_consume
        ctx.stats = fused_gt.gt_stats_pair_chosen(row_ptr, col_ind, val, Q)   # the block plan, or None
        if ctx.stats is not None:
            out_feat, row_max, row_sum = fused_gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V, ctx.stats, val)
            ctx.save_for_backward(row_ptr, col_ind, Q, K, V, row_max, row_sum, val)
            return out_feat
        ctx.ranked = fused_gt.gt_ranked_pair_chosen(row_ptr, col_ind, val, Q)   # the block plan, or None
        if ctx.ranked is not None:   # one head, all dense, unit values: attn_edge in rank order (same pair, cheaper forward)
            out_feat, attn_ranked = fused_gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V, ctx.ranked)
            ctx.save_for_backward(row_ptr, col_ind, Q, K, V, attn_ranked)
            return out_feat
        out_feat, attn_edge = fused_gt.gt_hyper_forward(
            row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem_consume, Q, K, V)
        ctx.save_for_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, Q, K, V, attn_edge)
        return out_feat

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.stats is not None:
            row_ptr, col_ind, Q, K, V, row_max, row_sum, val = ctx.saved_tensors
            grad_Q, grad_K, grad_V = fused_gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, row_max, row_sum,
                                                                grad_out.contiguous(), ctx.stats, val)
            return (None,) * 8 + (grad_Q, grad_K, grad_V)
        if ctx.ranked is not None:
            row_ptr, col_ind, Q, K, V, attn_ranked = ctx.saved_tensors
            grad_Q, grad_K, grad_V = fused_gt.gt_backward_ranked(row_ptr, col_ind, Q, K, V, attn_ranked,
                                                                 grad_out.contiguous(), ctx.ranked)
            return (None,) * 8 + (grad_Q, grad_K, grad_V)
        row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, Q, K, V, attn_edge = ctx.saved_tensors
        grad_Q, grad_K, grad_V = fused_gt.gt_backward(
            row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, ctx.smem, Q, K, V, attn_edge,
            grad_out.contiguous())
        return (None,) * 8 + (grad_Q, grad_K, grad_V)


def GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume, Q, K, V):
    """Differentiable hyper conv.  reference :51-76"""
    return FusedGTFunction_hyper.apply(
        rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume, Q, K, V)


def GTConvFuse_inference_softmax(indptr, indices, rows, val, smem_consume, Q, K, V):
    """softmax: two kernels (COO SDDMM, then softmax + SpMM).  reference :238-259"""
    return fused_gt.gt_softmax_inference(indptr, indices, rows, val, smem_consume, Q, K, V)[0]


def GTConvFuse_inference_softmax_gm(indptr, indices, rows, val, Q, K, V):
    """softmax with logits kept in global memory; bare tensor.  reference :262-279"""
    return fused_gt.gt_softmax_gm_inference(indptr, indices, rows, val, Q, K, V)


def GTConvFuse_inference_csr(indptr, indices, val, smem_consume, Q, K, V):
    """reference :282-301"""
    return fused_gt.gt_csr_inference(indptr, indices, val, smem_consume, Q, K, V)[0]


def GTConvFuse_inference_csr_gm(indptr, indices, val, Q, K, V):
    """reference :304-321"""
    return fused_gt.gt_csr_gm_inference(indptr, indices, val, Q, K, V)[0]


def GTConvFuse_inference_tiling(indptr, indices, val, smem_consume, Q, K, V):
    """tiling: one kernel, column tiles + online softmax.  reference :324-343"""
    return fused_gt.gt_tiling_inference(indptr, indices, val, smem_consume, Q, K, V)[0]
