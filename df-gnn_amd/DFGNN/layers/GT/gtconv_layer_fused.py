"""Inference layers for the hyper / softmax / csr formats (reference: DFGNN/layers/GT/gtconv_layer_fused.py:11-93).
forward(params, h, fuse=False) -> (out[N, dim], elapsed_ms)."""
from DFGNN.operators.fused_gtconv import (GTConvFuse_inference_csr, GTConvFuse_inference_hyper,
                                          GTConvFuse_inference_softmax)

from .gtconv_layer import SparseMHA


class SparseMHA_hyper(SparseMHA):
    def forward(self, params, h, fuse=False):
        if not fuse:
            return self._baseline(params, h)
        indptr, indices, rows, val, smem_consume = params
        return self._fused_inference(GTConvFuse_inference_hyper, (indptr, indices, rows, val, smem_consume), h)


class SparseMHA_CSR(SparseMHA):
    def forward(self, params, h, fuse=False):
        if not fuse:
            return self._baseline(params, h)
        indptr, indices, val, smem_consume = params
        return self._fused_inference(GTConvFuse_inference_csr, (indptr, indices, val, smem_consume), h)


class SparseMHA_softmax(SparseMHA):
    def forward(self, params, h, fuse=False):
        if not fuse:
            return self._baseline(params, h)
        indptr, indices, rows, val, smem_consume = params
        return self._fused_inference(GTConvFuse_inference_softmax, (indptr, indices, rows, val, smem_consume), h)
