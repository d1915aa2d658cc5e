#!/usr/bin/env python3
"""Per-graph max error of the matrix-core GT kernels on the batch of tests/test_gpu_parity.py::test_dense_kernels_every_geometry
(which graph / geometry is off when that test fails).  usage: dense_geom_probe.py H F"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt, oracle
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import Graph, batch, synthetic as S
oracle.build()
h, f = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
rng = np.random.default_rng(17 + f)
def er(n, p, drop_rows=()):
    iu, ju = np.triu_indices(n, k=1)
    keep = rng.random(len(iu)) < p
    keep &= ~np.isin(iu, drop_rows) & ~np.isin(ju, drop_rows)
    return np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]])
graphs, sizes = [], []
for n, p in ((9, 0.9), (17, 0.6), (64, 0.5), (128, 0.35), (129, 0.3), (145, 0.4), (160, 0.3), (161, 0.3), (200, 0.25), (255, 0.2)):
    s_, d_ = er(n, p); graphs.append(Graph(s_, d_, n)); sizes.append(n)
s_, d_ = er(70, 0.5, drop_rows=(0, 33, 69)); graphs.append(Graph(s_, d_, 70)); sizes.append(70)
keep = rng.random((80, 140)) < 0.3
ds_, dd_ = np.nonzero(keep); graphs.append(Graph(ds_.astype(np.int64), dd_.astype(np.int64), 140)); sizes.append(140)
s_, d_ = er(40, 0.6); graphs.append(Graph(np.concatenate([s_, s_[:1]]), np.concatenate([d_, d_[:1]]), 40)); sizes.append(40)
g = batch(graphs).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, h, f, seed=5, device=dev)
dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(dev)
args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
out, attn = gt.gt_hyper_forward(*args)
dQ, dK, dV = gt.gt_backward(*args, attn, dO)
n_ = lambda t: t.detach().cpu().numpy()
want = oracle.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V))
wq, wk, wv = oracle.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
off = np.concatenate([[0], np.cumsum(sizes)])
for k, n in enumerate(sizes):
    sl = slice(off[k], off[k + 1])
    e = [float(np.abs(n_(a)[sl].astype(np.float64) - b[sl]).max()) for a, b in ((out, want), (dQ, wq), (dK, wk), (dV, wv))]
    per_head = [float(np.abs(n_(dQ)[sl, hh].astype(np.float64) - wq[sl, hh]).max()) for hh in range(h)]
    print(f"graph {k:2d} n={n:3d}: out {e[0]:.2e} dQ {e[1]:.2e} dK {e[2]:.2e} dV {e[3]:.2e}   dQ per head " + " ".join(f"{x:.1e}" for x in per_head))
    if e[1] > 1e-3:
        bad = np.abs(n_(dQ)[sl].astype(np.float64) - wq[sl]).max(axis=(1, 2)) > 1e-3
        print("    bad rows:", np.nonzero(bad)[0][:40], "...", int(bad.sum()), "of", n)
