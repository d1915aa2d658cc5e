"""GAT operator surface -- same names / signatures as the reference's DFGNN/operators/fused_gatconv.py.

In scope: the hyper / softmax / softmax_gm / tiling inference functions and the training
pair `GATConvFuse` / `FusedGATFunction` and the hyper_v2 / hyper_recompute / hyper_ablation
variants of the reference's comparison sweeps.
"""
import fused_gatconv as fused_gat
import torch


def GATConvFuse_inference_hyper(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """reference :31-36"""
    return fused_gat.gat_inference_hyper(
        smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat)


def GATConvFuse_inference_hyper_ablation(smem_consume, attn_row, attn_col, indptr, indices, rows,
                                         negative_slope, in_feat):
    """reference :55-60"""
    return fused_gat.gat_inference_hyper_ablation(
        smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat)


def GATConvFuse_inference_softmax(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """reference :63-68"""
    return fused_gat.gat_inference_softmax(
        smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat)


def GATConvFuse_inference_softmax_gm(attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """reference :71-76"""
    return fused_gat.gat_inference_softmax_gm(attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat)


def GATConvFuse_inference_tiling(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """reference :79-84"""
    return fused_gat.gat_inference_tiling(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat)


def GATConvFuse_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """reference :87-92"""
    return fused_gat.gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat)


def GATConvFuse_inference_hyper_recompute(attn_row, attn_col, indptr, indices, negative_slope, in_feat):
    """reference :39-44"""
    return fused_gat.gat_inference_hyper_recompute(attn_row, attn_col, indptr, indices, negative_slope, in_feat)


def GATConvFuse_inference_hyper_v2(smem_consume, a_l, a_r, indptr, indices, negative_slope, in_feat):
    """reference :47-52"""
    return fused_gat.gat_inference_hyper_v2(smem_consume, a_l, a_r, indptr, indices, negative_slope, in_feat)


class FusedGATFunction(torch.autograd.Function):
    """reference :95-176 (training pair: fwd saves the row statistics, bwd recomputes the attention)."""

    @staticmethod
    def forward(ctx, attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute, negative_slope, in_feat,
                attn_drop):
        out_feat, edge_max, edge_sum, edge_mask = fused_gat.gat_forward(
            attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop)
        ctx.save_for_backward(row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask,
                              in_feat, attn_row, attn_col)
        ctx.negative_slope, ctx.attn_drop = negative_slope, attn_drop
        return out_feat

    @staticmethod
    def backward(ctx, grad_out):
        (row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, in_feat, attn_row,
         attn_col) = ctx.saved_tensors
        grad_feat, grad_attn_row, grad_attn_col = fused_gat.gat_backward(
            ctx.negative_slope, ctx.attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
            edge_mask, in_feat, attn_row, attn_col, grad_out.contiguous())
        return grad_attn_row, grad_attn_col, None, None, None, None, None, None, grad_feat, None


def GATConvFuse(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute, negative_slope, in_feat,
                attn_drop):
    """reference :5-28"""
    return FusedGATFunction.apply(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute,
                                  negative_slope, in_feat, attn_drop)
