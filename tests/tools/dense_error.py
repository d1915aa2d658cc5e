#!/usr/bin/env python3
"""Max abs / relative error of the matrix-core path vs the f64 oracle on a PATTERN-like sample (GT fwd, bwd; GAT)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt, fused_gatconv as gat, oracle
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
oracle.build()
dev = "cuda:0"
g = S.pattern_like(batch_size=96, seed=4).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, 1, 128, seed=11, device=dev)
dO = torch.randn(Q.shape, generator=torch.Generator().manual_seed(2)).to(dev)
n = lambda t: t.detach().cpu().numpy()
args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
want, wattn = oracle.gt_forward(n(row_ptr), n(col_ind), n(val), n(Q), n(K), n(V), want_attn=True)
wq, wk, wv = oracle.gt_backward(n(row_ptr), n(col_ind), n(val), n(Q), n(K), n(V), n(dO))
# DFGNN_DENSE=0 in the environment keeps every range on the edge-walking kernels (read once by the library)
if True:
    out, attn = gt.gt_hyper_forward(*args)
    dQ, dK, dV = gt.gt_backward(*args, attn, dO)
    print("plan num_dense", row_ptr._dfgnn_plans[128].num_dense, " DFGNN_DENSE =", os.environ.get("DFGNN_DENSE", "(unset: on)"))
    for name, a, b in (("out", out, want), ("attn", attn, wattn), ("dQ", dQ, wq), ("dK", dK, wk), ("dV", dV, wv)):
        e = np.abs(n(a).astype(np.float64) - b)
        print(f"  {name:5s} max abs err {e.max():.3e}   max |ref| {np.abs(b).max():.3e}   max err / (1e-3 + 1e-3 |ref|) {(e / (1e-3 + 1e-3 * np.abs(b))).max():.4f}")
ar, ac, X = S.gat_features(m, 1, 128, seed=6, device=dev)
wg = oracle.gat_forward(n(row_ptr), n(col_ind), n(ar), n(ac), 0.2, n(X))
e = np.abs(n(gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X)).astype(np.float64) - wg)
print(f"  GAT   max abs err {e.max():.3e}   max |ref| {np.abs(wg).max():.3e}")
