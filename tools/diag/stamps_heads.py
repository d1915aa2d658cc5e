#!/usr/bin/env python3
"""Diagnostic (needs the -DDFGNN_STAMPS build): phase times of the grouped-heads forward (dfgnn_dense_heads.hpp), wave 0,
first head of the first group.  usage: python3 tools/diag/stamps_heads.py [heads]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import dfgnn_native
dfgnn_native.LIB_PATH = os.path.join(ROOT, "df-gnn_amd", os.environ.get("DFGNN_STAMP_LIB", "libdfgnn_stamps.so"))
os.environ["DFGNN_BINDING"] = "ctypes"
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
h = int(sys.argv[1]) if len(sys.argv) > 1 else 8
f = 128 // h
g = S.pattern_like(batch_size=1024, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, h, f, seed=100, device=dev)
L = dfgnn_native.lib()
L.dfgnn_debug_set_dense_stamps.argtypes = [ctypes.c_void_p]
args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
for _ in range(3):
    out, attn = gt.gt_hyper_forward(*args)
nd = row_ptr._dfgnn_plans[f].num_dense
st = torch.zeros(nd * 16, dtype=torch.int64, device=dev)
assert L.dfgnn_debug_set_dense_stamps(st.data_ptr()) == 0
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record(); gt.gt_hyper_forward(*args); e1.record(); torch.cuda.synchronize()
assert L.dfgnn_debug_set_dense_stamps(None) == 0
s = st.cpu().numpy().reshape(nd, 16)
n = s[:, 15] >> 32
names = ["prologue+images", "head0: Q conv", "S", "softmax", "attn out", "PV", "rest of group 0", "store rows", "next images", "other groups"]
t = s[:, :11].astype(np.float64)
d = np.diff(t, axis=1)
print(f"heads {h} f {f}: {e0.elapsed_time(e1) * 1e3:.1f} us (stamped)")
for sel, lab in ((n <= 128, "n<=128"), ((n > 128) & (n <= 160), "129..160")):
    tot = t[sel, 10] - t[sel, 0]
    print(f"  {lab}: {sel.sum()} ranges, cycles per WG mean {tot.mean():.0f}")
    print("    " + "  ".join(f"{nm} {np.median(d[sel, k]):.0f}" for k, nm in enumerate(names)))
