#!/usr/bin/env python3
"""Kernel time of both GT training pairs (attn_edge / row statistics) on batches of equal-sized graphs, per range class.
usage: class_bench_stats.py [--heads H] [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
argv = sys.argv[1:]
heads = 1
if "--heads" in argv:
    k = argv.index("--heads"); heads = int(argv[k + 1]); del argv[k:k + 2]
sizes = [int(a) for a in argv] or [107, 128, 140, 160, 175]
bs, reps, dim = 1024, 20, 128
f = dim // heads
for n in sizes:
    g = S.pattern_like(batch_size=bs, seed=1, mean_nodes=float(n), std_nodes=0.0, lo=n, hi=n, mean_deg=0.43 * (n - 1)).to(dev)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, heads, f, seed=100, device=dev)
    dO = torch.randn_like(Q)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    o2, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
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
    sf = ev(lambda: gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V))
    sb = ev(lambda: gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO))
    kc = lambda t: t * 1e-6 * 2.2e9 * 256 / bs / 1e3
    print(f"h={heads} n={n:3d}: attn pair fwd {tf:6.1f} bwd {tb:6.1f} us | stats pair fwd {sf:6.1f} ({kc(sf):5.1f} kcyc/range) "
          f"bwd {sb:6.1f} ({kc(sb):5.1f} kcyc/range)", flush=True)
