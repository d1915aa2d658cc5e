"""SparseMHA base: Q/K/V projections + the non-fused baseline branch.
Same class / method names and tensor layouts as the reference's DFGNN/layers/GT/gtconv_layer.py:5-33;
`forward_dglsp` runs on DFGNN.utils.sparse (torch) because dgl.sparse does not exist on ROCm boxes."""
import torch.nn as nn

from DFGNN.utils import sparse as dglsp


class SparseMHA(nn.Module):
    """Sparse multi-head attention: out = softmax_rows((Q K^T) * A) V."""

    def __init__(self, in_size, out_size, num_heads):
        super().__init__()
        self.in_size = in_size
        self.num_heads = num_heads
        self.head_dim = out_size // num_heads
        self.scaling = self.head_dim ** -0.5
        self.q_proj = nn.Linear(in_size, out_size)
        self.k_proj = nn.Linear(in_size, out_size)
        self.v_proj = nn.Linear(in_size, out_size)

    def prep_qkv(self, h):
        """[N, head_dim, heads] layout (inference path), q pre-scaled.  reference :19-27"""
        N = len(h)
        q = self.q_proj(h).reshape(N, self.head_dim, self.num_heads) * self.scaling
        k = self.k_proj(h).reshape(N, self.head_dim, self.num_heads)
        v = self.v_proj(h).reshape(N, self.head_dim, self.num_heads)
        return q, k, v

    def forward_dglsp(self, A, q, k, v):
        attn = dglsp.bsddmm(A, q, k.transpose(1, 0))  # [nnz, nh]
        attn = attn.softmax()
        return dglsp.bspmm(attn, v)

    def _fused_inference(self, op, op_args, h):
        """Shared body of the inference layers: [N,d,nh] -> [N,nh,d] contiguous, timed fused op, back."""
        from DFGNN.utils import benchmark
        q, k, v = (t.transpose(1, 2).contiguous() for t in self.prep_qkv(h))
        out, elapsed = benchmark(op, *op_args, q, k, v)
        return out.transpose(1, 2).reshape(len(h), -1), elapsed * 1000

    def _baseline(self, A, h):
        from DFGNN.utils import benchmark
        q, k, v = self.prep_qkv(h)
        out, elapsed = benchmark(self.forward_dglsp, A, q, k, v)
        return out.reshape(len(h), -1), elapsed * 1000
