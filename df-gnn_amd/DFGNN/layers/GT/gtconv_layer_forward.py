"""Training layers: fused forward + fused backward through GTConvFuse_hyper
(reference: DFGNN/layers/GT/gtconv_layer_forward.py:7-104).

Layout note kept from the reference: the fused branch reshapes straight to
[N, heads, head_dim] while the baseline branch uses [N, head_dim, heads]; the two agree for heads == 1.
params = (A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume)."""
from DFGNN.operators.fused_gtconv import GTConvFuse_hyper, GTConvFuse_inference_hyper
from DFGNN.utils import benchmark

from .gtconv_layer import SparseMHA


class _TrainingQKV(SparseMHA):
    def _qkv_fused(self, h):
        N = len(h)
        q = self.q_proj(h).reshape(N, self.num_heads, self.head_dim) * self.scaling
        k = self.k_proj(h).reshape(N, self.num_heads, self.head_dim)
        v = self.v_proj(h).reshape(N, self.num_heads, self.head_dim)
        return q, k, v


class SparseMHA_forward(_TrainingQKV):
    def forward(self, params, h, fuse=False):
        A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume = params
        if fuse:
            q, k, v = self._qkv_fused(h)
            if self.training:
                out = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume,
                                       q.contiguous(), k.contiguous(), v.contiguous())
            else:
                out = GTConvFuse_inference_hyper(row_ptr, col_ind, rows, val, smem_consume, q.contiguous(),
                                                 k.contiguous(), v.contiguous())
        else:
            out = self.forward_dglsp(A, *self.prep_qkv(h))
        return out.reshape(len(h), -1)


class SparseMHA_forward_timing(_TrainingQKV):
    def forward(self, params, h, fuse=False):
        A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume = params
        if fuse:
            q, k, v = self._qkv_fused(h)
            out, elapsed = benchmark(GTConvFuse_hyper, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx,
                                     smem_consume, q.contiguous(), k.contiguous(), v.contiguous())
            out = out.transpose(1, 2)
        else:
            out, elapsed = benchmark(self.forward_dglsp, A, *self.prep_qkv(h))
        return out.reshape(len(h), -1), elapsed * 1000
