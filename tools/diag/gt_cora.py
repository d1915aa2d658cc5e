#!/usr/bin/env python3
"""Diagnostic: GT entry points on the cora-like graph (average degree 3.9, hub rows up to 168 edges): the low-degree
lane-group kernels ('hyper', nnz < 8 m) next to the wave-per-row 'tiling' kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import fused_gtconv as gt  # noqa: E402
from DFGNN.layers import preprocess_Hyper_fw_bw  # noqa: E402
from DFGNN.utils import benchmark  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402

dev = "cuda:0"
g = S.cora_like().to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, 1, 128, seed=1, device=dev)
dO = torch.randn_like(Q)
o1, t1 = benchmark(lambda: gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0])
o2, t2 = benchmark(lambda: gt.gt_tiling_inference(row_ptr, col_ind, val, 128, Q, K, V)[0])
(o3, attn), t3 = benchmark(lambda: gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V))
_, t4 = benchmark(lambda: gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO))
print(f"cora-like GT f=128: hyper inference {t1*1e6:.1f} us, tiling inference {t2*1e6:.1f} us, hyper training fwd "
      f"{t3*1e6:.1f} us, backward {t4*1e6:.1f} us; max |hyper - tiling| {float((o1-o2).abs().max()):.2e}")
