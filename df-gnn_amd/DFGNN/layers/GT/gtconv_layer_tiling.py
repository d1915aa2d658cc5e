"""reference: DFGNN/layers/GT/gtconv_layer_tiling.py"""
from DFGNN.operators.fused_gtconv import GTConvFuse_inference_tiling

from .gtconv_layer import SparseMHA


class SparseMHA_tiling(SparseMHA):
    def forward(self, params, h, fuse=False):
        if not fuse:
            return self._baseline(params, h)
        indptr, indices, val, smem_consume = params
        return self._fused_inference(GTConvFuse_inference_tiling, (indptr, indices, val, smem_consume), h)
