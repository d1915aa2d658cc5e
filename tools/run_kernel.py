#!/usr/bin/env python3
"""Profiling helper: run one hot-path entry point a few times on the C3 workload (for rocprofv3).
usage: python3 tools/run_kernel.py [fwd|bwd|fwd_stats|bwd_stats|pairs|step|fwd_infer|gat|gat_train|gat_train_drop|c4] [iters] [batch_size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import dfgnn_native  # noqa: E402
if os.environ.get("DFGNN_LIB"):  # A/B runs of two builds
    dfgnn_native.LIB_PATH = os.path.join(ROOT, "df-gnn_amd", os.environ["DFGNN_LIB"])
import fused_gtconv as gt  # noqa: E402
from DFGNN.layers import preprocess_Hyper_fw_bw  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = "cuda:0"
if what == "c4":  # BASELINE.json configs[3]: GAT 'tiling' on the reddit-like graph
    from DFGNN.layers.util import preprocess_CSR
    from DFGNN.operators import fused_gatconv as gatops
    g = S.reddit_like().to(dev)
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    ar, ac, X = S.gat_features(g.num_nodes(), 1, 128, seed=4, device=dev)
    for _ in range(iters):
        gatops.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)
    torch.cuda.synchronize()
    print("done c4", iters, g.num_nodes(), g.num_edges())
    sys.exit(0)
g = S.pattern_like(batch_size=bs, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, 1, 128, seed=100, device=dev)
dO = torch.randn_like(Q)
if what == "gat":
    import fused_gatconv as gat
    ar, ac, X = S.gat_features(m, 1, 128, seed=6, device=dev)
    for _ in range(iters):
        gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X)
    torch.cuda.synchronize()
    print("done gat", iters, m, g.num_edges())
    sys.exit(0)
if what.startswith("gat_train"):
    import fused_gatconv as gat
    ar, ac, X = S.gat_features(m, 1, 128, seed=6, device=dev)
    drop = 0.5 if what.endswith("drop") else 0.0  # (both take the matrix-core kernels on this batch)
    for _ in range(iters):
        out, emax, esum, mask = gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, drop)
        gat.gat_backward(0.2, drop, row_ptr, col_ind, col_ptr, row_ind, val_idx, emax, esum, mask, X, ar, ac, dO)
    torch.cuda.synchronize()
    print("done", what, iters, m, g.num_edges())
    sys.exit(0)
if what == "step":  # the headline step itself (autograd function: whatever pair it chooses), nothing else on the device
    from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
    Qg, Kg, Vg = (t.requires_grad_(True) for t in (Q, K, V))
    for _ in range(iters):
        o = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Qg, Kg, Vg)
        torch.autograd.grad(o, (Qg, Kg, Vg), dO)
    torch.cuda.synchronize()
    print("done", what, iters, m, g.num_edges())
    sys.exit(0)
if what == "pairs":  # the GT training pairs, fwd + bwd each, and the rank-ordered forward (one profiled process covers the five dense kernels)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    for _ in range(iters):
        out, attn = gt.gt_hyper_forward(*args)
        gt.gt_backward(*args, attn, dO)
        out, rmax, rsum = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
        gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, rmax, rsum, dO)
        out, attn_r = gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V)
    torch.cuda.synchronize()
    print("done", what, iters, m, g.num_edges())
    sys.exit(0)
if what.endswith("_stats"):  # the training pair the autograd functions take: (max, sum) per row instead of attn_edge
    assert gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q) is not None
    for _ in range(iters):
        out, rmax, rsum = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
        if what == "bwd_stats":
            gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, rmax, rsum, dO)
    torch.cuda.synchronize()
    print("done", what, iters, m, g.num_edges())
    sys.exit(0)
for _ in range(iters):
    if what == "fwd_infer":
        gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)
    else:
        out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        if what == "bwd":
            gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO)
torch.cuda.synchronize()
print("done", what, iters, m, g.num_edges())
