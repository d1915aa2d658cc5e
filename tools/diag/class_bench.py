#!/usr/bin/env python3
"""Kernel time of the 'hyper' GT forward / backward on batches of equal-sized graphs (one range class at a time), without
stamps: what a range of n nodes costs in the shipped build.  usage: class_bench.py [n ...]   (DFGNN_LIB=<file> for A/B)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
sizes = [int(a) for a in sys.argv[1:]] or [107, 128, 140, 160, 175]
bs, reps = 1024, 20
for n in sizes:
    g = S.pattern_like(batch_size=bs, seed=1, mean_nodes=float(n), std_nodes=0.0, lo=n, hi=n, mean_deg=0.43 * (n - 1)).to(dev)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, 1, 128, seed=100, device=dev)
    dO = torch.randn_like(Q)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    def ev(fn):
        for _ in range(3):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
    tf = ev(lambda: gt.gt_hyper_forward(*args))
    tb = ev(lambda: gt.gt_backward(*args, attn, dO))
    fb = 16 * m * 128 + 6 * nnz                  # bytes the forward moves: Q K V out, 2-byte edge coordinates, attn
    bb = 28 * m * 128 + 6 * nnz                  # backward: Q K V dO dQ dK dV, coordinates, attn
    print(f"n={n:3d} (m={m}, nnz={nnz}): fwd {tf:6.1f} us = {tf*1e-6*2.2e9*256/bs/1e3:5.1f} kcyc/range@2.2GHz, {fb/tf/1e6:5.2f} TB/s | "
          f"bwd {tb:6.1f} us = {tb*1e-6*2.2e9*256/bs/1e3:5.1f} kcyc/range, {bb/tb/1e6:5.2f} TB/s")
