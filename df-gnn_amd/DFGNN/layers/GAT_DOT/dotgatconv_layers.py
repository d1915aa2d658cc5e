"""DOTGAT layers (dot-product graph attention) on the fused GT operators: K = V = H, the projected features, and
Q = H / sqrt(out_size).  Same class names, constructor and forward(params, feat, fuse) contract as the reference's
DFGNN/layers/GAT_DOT/*.py (dotgatconv_layer.py:5-16 base; dotgatconv_layer_csr.py:7-21, dotgatconv_layer_hyper.py:7-31,
dotgatconv_layer_softmax.py:8-32) -- no new kernels.  params = (g, *preprocess tuple): the graph
itself comes first, for the non-fused branch.

The reference's non-fused branch is `dgl.nn.DotGatConv` (dotgatconv_layer.py:1,12), a dependency this build does not
have; `DotGatConv` below restates what that module computes as DGL publishes it (`u_dot_v` -> logits / sqrt(out_feats)
-> `edge_softmax` over the edges that arrive at a node -> `u_mul_e` + `sum`) with torch index ops, same parameter (`fc`,
no bias) and output shape [N, heads, out].  PARITY WITH dgl UNPINNED: dgl is not importable here and the reference
holds no fixture of this module's output; the 1/sqrt(out_feats) factor in particular is from the published source, not
from a run.  It normalises over IN-edges while the fused operators normalise over a row's OUT-edges (rows = sources of
g.edges(), DFGNN/layers/util.py:53-56); the two agree on the symmetric graphs every dataset of the reference provides.

One deliberate difference from the reference's fused branch: it hands the operator (H, H, H)
(dotgatconv_layer_hyper.py:22-24), i.e. unscaled logits, which differs from its own non-fused branch by that factor;
here the fused branch scales Q so that fuse=True and fuse=False of one module agree (the layer's own check).  The
operator underneath is the same call with the same argument meaning."""
import torch
from torch import nn

from DFGNN.operators import fused_gtconv as ops


class DotGatConv(nn.Module):
    def __init__(self, in_feats, out_feats, num_heads):
        super().__init__()
        self._out_feats, self._num_heads = out_feats, num_heads
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)

    def forward(self, graph, feat):
        src, dst = graph.edges()
        n = graph.num_nodes()
        h = self.fc(feat).view(-1, self._num_heads, self._out_feats)
        a = (h[src] * h[dst]).sum(-1) / self._out_feats ** 0.5            # u_dot_v / sqrt(out_feats): [E, heads]
        amax = torch.full((n, self._num_heads), float("-inf"), device=a.device, dtype=a.dtype)
        amax = amax.scatter_reduce(0, dst[:, None].expand_as(a), a, reduce="amax", include_self=True)
        p = torch.exp(a - amax[dst])
        den = torch.zeros((n, self._num_heads), device=a.device, dtype=a.dtype).index_add_(0, dst, p)
        sa = p / den[dst]                                                 # edge_softmax over the in-edges of dst
        out = torch.zeros_like(h).index_add_(0, dst, h[src] * sa[:, :, None])
        return out


class DOTGATConvDGL(nn.Module):
    def __init__(self, in_size, out_size, num_heads):
        super().__init__()
        self.in_size, self.out_size, self.num_heads = in_size, out_size, num_heads
        self.conv_nofuse = DotGatConv(in_size, out_size, num_heads)

    def forward_dglsp(self, g, feat):
        return self.conv_nofuse(g, feat)


class _DOTGATInference(DOTGATConvDGL):
    """forward(params, feat, fuse) -> (out[N, heads * out], elapsed_ms), timed like the reference (3 dry + 10 runs)."""
    op = None

    def forward(self, params, feat, fuse=False):
        from DFGNN.utils import benchmark
        N = len(feat)
        g, graph_args = params[0], params[1:]
        if fuse:
            H = self.conv_nofuse.fc(feat).view(-1, self.num_heads, self.out_size).detach().contiguous()
            out, elapsed = benchmark(type(self).op, *graph_args, (H * self.out_size ** -0.5).contiguous(), H, H)
        else:
            with torch.no_grad():
                out, elapsed = benchmark(self.forward_dglsp, g, feat)
        return out.reshape(N, -1), elapsed * 1000


class DOTGATConv_csr(_DOTGATInference):        # params = (g, indptr, indices, val, smem)
    op = staticmethod(ops.GTConvFuse_inference_csr)


class DOTGATConv_hyper(_DOTGATInference):      # params = (g, indptr, indices, rows, val, smem)
    op = staticmethod(ops.GTConvFuse_inference_hyper)


class DOTGATConv_softmax(_DOTGATInference):    # params = (g, indptr, indices, rows, val, smem)
    op = staticmethod(ops.GTConvFuse_inference_softmax)
