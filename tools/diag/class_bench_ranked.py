#!/usr/bin/env python3
"""Kernel time of the rank-ordered training forward (and the statistics forward) on batches of equal-sized graphs.
usage: [DFGNN_LIB=libdfgnn_<variant>.so] class_bench_ranked.py [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
sizes = [int(a) for a in sys.argv[1:]] or [107, 128, 140, 160]
for n in sizes:
    g = S.pattern_like(batch_size=1024, seed=1, mean_nodes=float(n), std_nodes=0.0, lo=n, hi=n, mean_deg=0.43 * (n - 1)).to(dev)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    Q, K, V = S.gt_features(g.num_nodes(), 1, 128, seed=100, device=dev)
    def ev(fn, reps=20):
        for _ in range(3):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    print(f"n={n}: ranked fwd {ev(lambda: gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V)):6.1f}  "
          f"csr-order fwd {ev(lambda: gt.gt_hyper_forward(*args)):6.1f}  stats fwd {ev(lambda: gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)):6.1f} us", flush=True)
