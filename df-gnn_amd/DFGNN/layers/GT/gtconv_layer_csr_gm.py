"""reference: DFGNN/layers/GT/gtconv_layer_csr_gm.py"""
from DFGNN.operators.fused_gtconv import GTConvFuse_inference_csr_gm

from .gtconv_layer import SparseMHA


class SparseMHA_CSR_GM(SparseMHA):
    def forward(self, params, h, fuse=False):
        if not fuse:
            return self._baseline(params, h)
        indptr, indices, val, _ = params
        return self._fused_inference(GTConvFuse_inference_csr_gm, (indptr, indices, val), h)
