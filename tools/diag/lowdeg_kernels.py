#!/usr/bin/env python3
"""Diagnostic (for rocprofv3 --kernel-trace): the low-degree kernels on a Peptides-like batch and on the cora-like
graph, GT 'hyper' fwd+bwd and the GAT training pair, 20 launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fused_gatconv as gat  # noqa: E402
import fused_gtconv as gt  # noqa: E402
from DFGNN.layers import preprocess_Hyper_fw_bw  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402

dev = "cuda:0"
for name, g, h in (("peptides", S.peptides_like(batch_size=256, seed=3), 4), ("cora", S.cora_like(), 1)):
    g = g.to(dev)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, f = g.num_nodes(), 128 // h
    Q, K, V = S.gt_features(m, h, f, seed=3, device=dev)
    ar, ac, X = S.gat_features(m, h, f, seed=4, device=dev)
    dO = torch.randn_like(Q)
    torch.cuda.synchronize()
    marker = torch.zeros(1 if name == "peptides" else 2, device=dev)  # (a fill kernel separates the two graphs in the trace)
    for _ in range(20):
        out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO)
        o, emax, esum, mask = gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0)
        gat.gat_backward(0.2, 0.0, row_ptr, col_ind, col_ptr, row_ind, val_idx, emax, esum, mask, X, ar, ac, dO)
    torch.cuda.synchronize()
print("done")
