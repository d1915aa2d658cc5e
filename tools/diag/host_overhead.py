#!/usr/bin/env python3
"""Host time per operator call (no device sync inside the loop) and its profile: where a launch-bound call spends its
microseconds.  usage: host_overhead.py [gat_forward|gt_forward|gt_backward]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch
import fused_gatconv as gat, fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
what = sys.argv[1] if len(sys.argv) > 1 else "gat_forward"
dev = "cuda:0"
g = S.pattern_like(batch_size=64, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, 1, 128, seed=1, device=dev)
ar, ac, X = S.gat_features(m, 1, 128, seed=2, device=dev)
out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
fns = {"gat_forward": lambda: gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0),
       "gt_forward": lambda: gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V),
       "gt_backward": lambda: gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, out)}
fn = fns[what]
for _ in range(20):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    fn()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"{what}: {(t1 - t0) / 200 * 1e6:.1f} us of host time per call")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    fn()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
