"""reference: DFGNN/layers/GAT/gatconv_layer_tiling.py:7-34"""
from DFGNN.operators.fused_gatconv import GATConvFuse_inference_tiling

from .gatconv_layer import GATConvDGL


class GATConv_tiling(GATConvDGL):
    def conv(self, row_ptr, col_ind, a_l, a_r, h):
        attn_row, attn_col = self._attn_scores(a_l, a_r, h)
        return GATConvFuse_inference_tiling(attn_row, attn_col, row_ptr, col_ind, self.negative_slope, h)

    def forward(self, params, feat, fuse=False):
        if not fuse:
            return self._baseline(params, feat)
        row_ptr, col_ind, _, _ = params
        return self._fused((row_ptr, col_ind), feat)
