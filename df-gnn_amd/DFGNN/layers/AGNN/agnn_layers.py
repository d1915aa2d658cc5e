"""AGNN layers (attention = softmax over the cosine similarity of the projected features) on the fused GT operators:
Q = K = row-normalised H, V = H.  Same class names, constructor and forward(params, feat, fuse) contract as the
reference's DFGNN/layers/AGNN/*.py (agnn_layer.py:7-19 base; agnn_layer_fused.py:13-122 csr / softmax / hyper;
agnn_layer_tiling.py, agnn_layer_csr_gm.py, agnn_layer_softmax_gm.py; agnn_layer_forward.py:8-66 training) -- no new
kernels.  Like the reference, the projection is Linear(in, out) viewed as
[N, heads, out], i.e. the layers are single-head (heads = 1) in practice.  One table-driven body instead of the reference's per-format copies: a variant is the
operator it calls plus how it picks that operator's arguments out of the preprocess tuple."""
from torch import nn
from torch.nn import functional as F

from DFGNN.operators import fused_gtconv as ops
from DFGNN.utils import sparse as dglsp


class AGNNConvDGL(nn.Module):
    def __init__(self, in_size, out_size, num_heads):
        super().__init__()
        self.in_size, self.out_size, self.num_heads = in_size, out_size, num_heads
        self.proj = nn.Linear(in_size, out_size)

    def forward_dglsp(self, A, H):
        """Non-fused branch; H is [N, out, heads] (reference agnn_layer.py:14-19)."""
        H_norm = F.normalize(H, p=2, dim=1)
        attn = dglsp.bsddmm(A, H_norm, H_norm.transpose(1, 0)).softmax()
        return dglsp.bspmm(attn, H)


class _AGNNInference(AGNNConvDGL):
    """forward(params, feat, fuse) -> (out[N, heads * out], elapsed_ms), timed like the reference (3 dry + 10 runs)."""
    op = None            # fused operator: op(*graph_args, Q, K, V)
    n_params = 0         # length of the preprocess tuple this format takes

    def conv(self, H, *graph_args):
        H_norm = F.normalize(H, p=2, dim=-1)
        return type(self).op(*graph_args, H_norm, H_norm, H)

    def graph_args(self, params):
        return params

    def forward(self, params, feat, fuse=False):
        from DFGNN.utils import benchmark
        N = len(feat)
        H = self.proj(feat).view(-1, self.num_heads, self.out_size)
        if fuse:
            assert len(params) == self.n_params
            out, elapsed = benchmark(self.conv, H.detach().contiguous(), *self.graph_args(params))
        else:
            H = H.detach().reshape(-1, self.out_size, self.num_heads)
            out, elapsed = benchmark(self.forward_dglsp, params, H)
            out = out.transpose(1, 2)
        return out.reshape(N, -1), elapsed * 1000


class AGNNConv_csr(_AGNNInference):          # params = preprocess_CSR: (indptr, indices, val, smem)
    op, n_params = staticmethod(ops.GTConvFuse_inference_csr), 4


class AGNNConv_tiling(_AGNNInference):
    op, n_params = staticmethod(ops.GTConvFuse_inference_tiling), 4


class AGNNConv_csr_gm(_AGNNInference):       # the _gm operators take no smem_consume
    op, n_params = staticmethod(ops.GTConvFuse_inference_csr_gm), 4

    def graph_args(self, params):
        return params[:3]


class AGNNConv_hyper(_AGNNInference):        # params = preprocess_Hyper: (indptr, indices, rows, val, smem)
    op, n_params = staticmethod(ops.GTConvFuse_inference_hyper), 5


class AGNNConv_softmax(_AGNNInference):
    op, n_params = staticmethod(ops.GTConvFuse_inference_softmax), 5


class AGNNConv_softmax_gm(_AGNNInference):
    op, n_params = staticmethod(ops.GTConvFuse_inference_softmax_gm), 5

    def graph_args(self, params):
        return params[:4]


class AGNNConv_forward(AGNNConvDGL):
    """Training layer on the differentiable hyper operator; params = preprocess_Hyper_fw_bw's 9-tuple; returns the
    output only (reference agnn_layer_forward.py:38-66)."""

    def conv(self, H, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume):
        H_norm = F.normalize(H, p=2, dim=-1)
        return ops.GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem_consume, H_norm, H_norm, H)

    def forward(self, params, feat, fuse=False):
        N = len(feat)
        A, graph = params[0], params[1:]
        if fuse:
            out = self.conv(self.proj(feat).view(-1, self.num_heads, self.out_size), *graph)
        else:
            out = self.forward_dglsp(A, self.proj(feat).view(-1, self.out_size, self.num_heads))
        return out.reshape(N, -1)
