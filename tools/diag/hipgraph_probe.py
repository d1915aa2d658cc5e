#!/usr/bin/env python3
"""Diagnostic: which operator calls survive HIP-graph capture.  usage: hipgraph_probe.py <case>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fused_gtconv as gt  # noqa: E402
from DFGNN.layers import preprocess_Hyper_fw_bw  # noqa: E402
from DFGNN.utils import GraphedStep  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402

case = sys.argv[1]
dev = "cuda:0"
shape, what = case.split(":")
if shape == "pep":
    g, h, f = S.peptides_like(batch_size=32, seed=3).to(dev), 4, 32
elif shape == "pep1":
    g, h, f = S.peptides_like(batch_size=32, seed=3).to(dev), 1, 128
else:
    g, h, f = S.pattern_like(batch_size=24, seed=2).to(dev), 1, 128
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, h, f, seed=3, device=dev)
dO = torch.randn(m, h, f, device=dev)
out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
fns = {
    "fwd": lambda: gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V),
    "infer": lambda: gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V),
    "bwd": lambda: gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO),
    "tiling": lambda: gt.gt_tiling_inference(row_ptr, col_ind, val, 128, Q, K, V),
    "torch": lambda: (Q * 2.0).sum(),
    "both": lambda: gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V,
                                   gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K,
                                                       V)[1], dO),
}
if what == "autograd":
    from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
    Q, K, V = (t.requires_grad_(True) for t in (Q, K, V))

    def _step():
        o = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        return (o,) + torch.autograd.grad(o, (Q, K, V), dO)
    fns["autograd"] = _step
print(case, "m", m, "nnz", g.num_edges(), flush=True)
gs = GraphedStep(fns[what])
gs.replay()
torch.cuda.synchronize()
print(case, "OK", flush=True)
