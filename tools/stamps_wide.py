#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the matrix-core backward, by range class (needs the -DDFGNN_STAMPS build:
tools/diag/build_variant.sh stamps -DDFGNN_STAMPS)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import dfgnn_native
dfgnn_native.LIB_PATH = os.path.join(ROOT, "df-gnn_amd", os.environ.get("DFGNN_STAMP_LIB", "libdfgnn_stamps.so"))
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = S.pattern_like(batch_size=bs, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, 1, 128, seed=100, device=dev)
L = dfgnn_native.lib()
L.dfgnn_debug_set_dense_stamps.argtypes = [ctypes.c_void_p]
args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
for _ in range(3):
    out, attn = gt.gt_hyper_forward(*args)
nd = row_ptr._dfgnn_plans[128].num_dense
dO = torch.randn_like(out)
for _ in range(2):
    gt.gt_backward(*args, attn, dO)
st = torch.zeros(nd * 16, dtype=torch.int64, device=dev)
assert L.dfgnn_debug_set_dense_stamps(st.data_ptr()) == 0
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); gt.gt_backward(*args, attn, dO); e1.record(); torch.cuda.synchronize()
assert L.dfgnn_debug_set_dense_stamps(None) == 0
s = st.cpu().numpy().reshape(nd, 16)
n = s[:, 15] >> 32; ne = s[:, 15] & 0xffffffff
print("bwd ms", e0.elapsed_time(e1), " ranges", nd)
small = n <= 128
tot = (s[small, 8] - s[small, 0]).astype(np.float64)
print(f"n<=128: {small.sum()} ranges, nodes {n[small].mean():.0f}, edges {ne[small].mean():.0f}: cycles mean {tot.mean():.0f} p50 {np.median(tot):.0f} max {tot.max():.0f}")
wide = (n > 128) & (n <= 160)
if wide.sum():
    names = ["prologue(tile,dO.0)", "dV.0", "V.0 img", "dP.0", "dO.1+dV.1+V.1+dP.1", "t,dS", "dS->tile,K.0 img", "dQ.0", "Q.0 img", "dK.0", "K.1,dQ.1,Q.1,dK.1"]
    t = s[wide][:, :12].astype(np.float64)
    d = np.diff(t, axis=1)
    tot = t[:, 11] - t[:, 0]
    print(f"129..160: {wide.sum()} ranges, nodes {n[wide].mean():.0f}, edges {ne[wide].mean():.0f}: cycles mean {tot.mean():.0f} p50 {np.median(tot):.0f} max {tot.max():.0f}")
    print("   " + "  ".join(f"{nm} {np.median(d[:, k]):.0f}" for k, nm in enumerate(names)))
    w = s[wide].astype(np.float64)
    print("   V.0 turnover: post(wait+max)+barrier %.0f  convert+store %.0f  fetch issue %.0f  barrier %.0f" % (
        np.median(w[:, 12] - w[:, 2]), np.median(w[:, 13] - w[:, 12]), np.median(w[:, 14] - w[:, 13]), np.median(w[:, 3] - w[:, 14])))
big = n > 160
if big.sum():
    tot = (s[big, 8] - s[big, 0]).astype(np.float64)
    print(f">160: {big.sum()} ranges: cycles (last row block) mean {tot.mean():.0f}")
allt = np.where(wide, s[:, 11] - s[:, 0], s[:, 8] - s[:, 0]).astype(np.float64)
print("sum of WG cycles / 256 CUs:", allt.sum() / 256, " span:", (np.where(wide, s[:, 11], s[:, 8]).max() - s[:, 0].min()))
