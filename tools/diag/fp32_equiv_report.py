#!/usr/bin/env python3
"""Errors of the three GT training paths (attn_edge pair on the matrix cores, statistics pair, fp32 VALU kernels) against
the float64 oracle on the adversarial inputs of test_dense_kernels_are_fp32_equivalent, relative to the largest element."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt, oracle
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import Graph, batch
DEV = "cuda:0"
rng = np.random.default_rng(23)
graphs = []
for n, p in ((40, 0.5), (96, 0.4), (128, 0.4), (131, 0.4), (160, 0.3), (170, 0.3), (255, 0.15)):
    iu, ju = np.triu_indices(n, k=1); keep = rng.random(len(iu)) < p
    graphs.append(Graph(np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]]), n))
g = batch(graphs).to(DEV)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
n_ = lambda t: t.cpu().numpy()
for h, f in ((1, 128), (8, 16)):
    for sqk, sv, sdo in ((3.0, 1.0, 1.0), (1.0, 1e-6, 1e5), (0.05, 3e4, 1e-7)):
        gen = torch.Generator().manual_seed(7)
        Q = (torch.randn(m, h, f, generator=gen) * sqk).to(DEV); K = (torch.randn(m, h, f, generator=gen) * sqk).to(DEV)
        V = torch.randn(m, h, f, generator=gen) * sv; V[::3] *= 1e-3; V = V.to(DEV)
        dO = (torch.randn(m, h, f, generator=gen) * sdo).to(DEV)
        args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        o_a, attn = gt.gt_hyper_forward(*args); g_a = gt.gt_backward(*args, attn, dO)
        o_s, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V); g_s = gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)
        gt.USE_BLOCK_PLAN = False
        o_v, attn_v = gt.gt_hyper_forward(*args); g_v = gt.gt_backward(*args, attn_v, dO)
        gt.USE_BLOCK_PLAN = True
        want = oracle.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V))
        wg = oracle.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
        rel = lambda a, b: float(np.abs(n_(a).astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-300))
        print(f"h={h} f={f} scales {sqk} {sv} {sdo}:")
        for name, o, gr in (("attn pair ", o_a, g_a), ("stats pair", o_s, g_s), ("fp32 VALU ", o_v, g_v)):
            print(f"   {name}: out {rel(o, want):.2e}  dQ {rel(gr[0], wg[0]):.2e}  dK {rel(gr[1], wg[1]):.2e}  dV {rel(gr[2], wg[2]):.2e}", flush=True)
