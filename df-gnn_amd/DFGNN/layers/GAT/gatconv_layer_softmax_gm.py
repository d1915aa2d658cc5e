"""reference: DFGNN/layers/GAT/gatconv_layer_softmax_gm.py"""
from DFGNN.operators.fused_gatconv import GATConvFuse_inference_softmax_gm

from .gatconv_layer import GATConvDGL


class GATConv_softmax_gm(GATConvDGL):
    def conv(self, indptr, indices, rows, a_l, a_r, h):
        attn_row, attn_col = self._attn_scores(a_l, a_r, h)
        return GATConvFuse_inference_softmax_gm(attn_row, attn_col, indptr, indices, rows, self.negative_slope, h)

    def forward(self, params, feat, fuse=False):
        if not fuse:
            return self._baseline(params, feat)
        indptr, indices, rows, _, _ = params
        return self._fused((indptr, indices, rows), feat)
