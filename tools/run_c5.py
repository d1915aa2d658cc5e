#!/usr/bin/env python3
"""Profiling helper: C5 (Peptides-like, multi-head) fwd+bwd steps for rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
heads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
g = S.peptides_like(batch_size=256, seed=3).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes(); f = 128 // heads
Q, K, V = S.gt_features(m, heads, f, seed=3, device=dev)
dO = torch.randn_like(Q)
for _ in range(10):
    out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO)
torch.cuda.synchronize()
print("done", m, g.num_edges())
