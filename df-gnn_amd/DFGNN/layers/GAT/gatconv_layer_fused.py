"""GAT inference layers, hyper / softmax / dgNN-csr formats
(reference: DFGNN/layers/GAT/gatconv_layer_fused.py:11-156).
forward(params, feat, fuse=False) -> (out[N, heads*out], elapsed_ms)."""
from DFGNN.operators.fused_gatconv import (GATConvFuse_inference, GATConvFuse_inference_hyper,
                                           GATConvFuse_inference_hyper_ablation,
                                           GATConvFuse_inference_hyper_recompute, GATConvFuse_inference_hyper_v2,
                                           GATConvFuse_inference_softmax)

from .gatconv_layer import GATConvDGL


class GATConv_dgNN(GATConvDGL):
    def conv(self, row_ptr, col_ind, a_l, a_r, h):
        attn_row, attn_col = self._attn_scores(a_l, a_r, h)
        return GATConvFuse_inference(attn_row, attn_col, row_ptr, col_ind, self.negative_slope, h)

    def forward(self, params, feat, fuse=False):
        if not fuse:
            return self._baseline(params, feat)
        row_ptr, col_ind, _, _ = params
        return self._fused((row_ptr, col_ind), feat)


class GATConv_hyper(GATConvDGL):
    def conv(self, indptr, indices, rows, smem_consume, a_l, a_r, h):
        attn_row, attn_col = self._attn_scores(a_l, a_r, h)
        return GATConvFuse_inference_hyper(smem_consume, attn_row, attn_col, indptr, indices, rows,
                                           self.negative_slope, h)

    def forward(self, params, feat, fuse=False):
        if not fuse:
            return self._baseline(params, feat)
        indptr, indices, rows, _, smem_consume = params
        return self._fused((indptr, indices, rows, smem_consume), feat)


class GATConv_softmax(GATConvDGL):
    def conv(self, indptr, indices, rows, smem_consume, a_l, a_r, h):
        attn_row, attn_col = self._attn_scores(a_l, a_r, h)
        return GATConvFuse_inference_softmax(smem_consume, attn_row, attn_col, indptr, indices, rows,
                                             self.negative_slope, h)

    def forward(self, params, feat, fuse=False):
        if not fuse:
            return self._baseline(params, feat)
        indptr, indices, rows, _, smem_consume = params
        return self._fused((indptr, indices, rows, smem_consume), feat)


class GATConv_hyper_ablation(GATConv_hyper):
    """reference :119-156 (ablation study entry; served by the production kernel)."""

    def conv(self, indptr, indices, rows, smem_consume, a_l, a_r, h):
        attn_row, attn_col = self._attn_scores(a_l, a_r, h)
        return GATConvFuse_inference_hyper_ablation(smem_consume, attn_row, attn_col, indptr, indices, rows,
                                                    self.negative_slope, h)


class GATConv_hyper_recompute(GATConvDGL):
    """reference :85-117: CSR arrays of the hyper tuple, logits recomputed instead of stored."""

    def conv(self, indptr, indices, a_l, a_r, h):
        attn_row, attn_col = self._attn_scores(a_l, a_r, h)
        return GATConvFuse_inference_hyper_recompute(attn_row, attn_col, indptr, indices, self.negative_slope, h)

    def forward(self, params, feat, fuse=False):
        if not fuse:
            return self._baseline(params, feat)
        indptr, indices, _, _, _ = params
        return self._fused((indptr, indices), feat)


class GATConv_hyper_v2(GATConvDGL):
    """reference :159-190: the attention scores are computed inside the operator (one pass over the features)."""

    def conv(self, indptr, indices, smem_consume, a_l, a_r, h):
        return GATConvFuse_inference_hyper_v2(smem_consume, a_l, a_r, indptr, indices, self.negative_slope, h)

    def forward(self, params, feat, fuse=False):
        if not fuse:
            return self._baseline(params, feat)
        indptr, indices, _, _, smem_consume = params
        return self._fused((indptr, indices, smem_consume), feat)
