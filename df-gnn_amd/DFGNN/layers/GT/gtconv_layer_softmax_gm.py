"""reference: DFGNN/layers/GT/gtconv_layer_softmax_gm.py"""
from DFGNN.operators.fused_gtconv import GTConvFuse_inference_softmax_gm

from .gtconv_layer import SparseMHA


class SparseMHA_softmax_gm(SparseMHA):
    def forward(self, params, h, fuse=False):
        if not fuse:
            return self._baseline(params, h)
        indptr, indices, rows, val, _ = params
        return self._fused_inference(GTConvFuse_inference_softmax_gm, (indptr, indices, rows, val), h)
