"""GATConvDGL base: W projection, a_l / a_r attention vectors and the non-fused baseline branch.
Same names, parameter shapes and init as the reference's DFGNN/layers/GAT/gatconv_layer.py:6-38;
`forward_dglsp` runs on DFGNN.utils.sparse (torch) instead of dgl.sparse."""
import torch
from torch import nn

from DFGNN.utils import sparse as dglsp


class GATConvDGL(nn.Module):
    def __init__(self, in_size, out_size, num_heads, dropout=0, negative_slope=0.2):
        super().__init__()
        self.in_size = in_size
        self.out_size = out_size
        self.num_heads = num_heads
        self.negative_slope = negative_slope
        self.dropout = nn.Dropout(dropout)
        self.W = nn.Linear(in_size, out_size * num_heads)
        self.a_l = nn.Parameter(torch.zeros(1, out_size, num_heads))
        self.a_r = nn.Parameter(torch.zeros(1, out_size, num_heads))
        self.activation = nn.LeakyReLU(negative_slope=negative_slope)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.W.weight, gain=gain)
        nn.init.xavier_normal_(self.a_l, gain=gain)
        nn.init.xavier_normal_(self.a_r, gain=gain)

    def forward_dglsp(self, A_hat, Z):
        # a^T [Wh_i || Wh_j] = a_l Wh_i + a_r Wh_j ; Z is [N, out, heads]
        e_l = (Z * self.a_l).sum(dim=1)
        e_r = (Z * self.a_r).sum(dim=1)
        a = self.activation(e_l[A_hat.row] + e_r[A_hat.col])
        return dglsp.bspmm(dglsp.val_like(A_hat, a).softmax(), Z)

    # ---- shared bodies of the fused / baseline branches -------------------------------------
    def _attn_scores(self, a_l, a_r, h):
        """attn_row/attn_col [N, heads]; timed together with the kernel, as in the reference
        (SURVEY.md 9 #8)."""
        return (a_l * h).sum(dim=-1), (a_r * h).sum(dim=-1)

    def _fused(self, conv_args, feat):
        from DFGNN.utils import benchmark
        N = len(feat)
        feat = self.W(feat).view(-1, self.num_heads, self.out_size).detach().contiguous()
        out, elapsed = benchmark(self.conv, *conv_args, self.a_l.transpose(1, 2).detach(),
                                 self.a_r.transpose(1, 2).detach(), feat)
        return out.reshape(N, -1), elapsed * 1000

    def _baseline(self, A, feat):
        from DFGNN.utils import benchmark
        N = len(feat)
        feat = self.W(feat).view(-1, self.out_size, self.num_heads).detach().contiguous()
        out, elapsed = benchmark(self.forward_dglsp, A, feat)
        return out.reshape(N, -1), elapsed * 1000
