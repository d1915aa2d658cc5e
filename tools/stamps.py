#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the block kernel (needs the -DDFGNN_STAMPS build)."""
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
for _ in range(3):
    gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
nfit = row_ptr._dfgnn_plans[128].num_fit
st = torch.zeros(nfit * 8, dtype=torch.int64, device=dev)
L.dfgnn_debug_set_stamps.argtypes = [ctypes.c_void_p]
assert L.dfgnn_debug_set_stamps(st.data_ptr()) == 0
rs = torch.zeros(nfit * 8, dtype=torch.int64, device=dev)
if hasattr(L, "dfgnn_debug_set_round_stamps"):
    L.dfgnn_debug_set_round_stamps.argtypes = [ctypes.c_void_p]
    L.dfgnn_debug_set_round_stamps(rs.data_ptr())
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V); e1.record()
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(nfit, 8)
t = s[:, :7].astype(np.float64)
n = s[:, 7] >> 32; ne = s[:, 7] & 0xffffffff
d = np.diff(t, axis=1)
names = ["load K+idx", "pass A", "barrier wait", "load V", "pass B", "(end)"]
print("kernel ms", e0.elapsed_time(e1), "nfit", nfit, "plan", row_ptr._dfgnn_plans[128].meta)
for k, nm in enumerate(names[:5]):
    print(f"{nm:14s} mean {d[:,k].mean():9.0f} ticks  p50 {np.median(d[:,k]):9.0f}  max {d[:,k].max():9.0f}")
tot = t[:, 5] - t[:, 0]
print("total per WG mean", tot.mean(), "max", tot.max(), "min", tot.min())
print("span of kernel (ticks):", t[:, 5].max() - t[:, 0].min())
print("sum of WG totals / 256 CUs:", tot.sum() / 256)
rr = rs.cpu().numpy().reshape(nfit, 8).astype(np.float64)
if rr[:, 5].sum() > 0:
    print("MFMA rounds per WG mean", rr[:, 5].mean(), " per-WG cycle sums: split+prefetch %.0f  tiles %.0f  barrier1 %.0f  gather %.0f  barrier2 %.0f"
          % tuple(rr[:, k].mean() for k in range(5)))
print("corr(total, ne):", np.corrcoef(tot, ne)[0, 1], " ticks per edge:", (tot / np.maximum(ne, 1)).mean())
print("nodes: mean", n.mean(), "max", n.max(), " edges mean", ne.mean(), "max", ne.max())
order = np.argsort(t[:, 0]); starts = t[order, 0] - t[:, 0].min()
print("start times of WGs #0,255,256,511,512,768,last:", [int(starts[i]) for i in (0, min(255,nfit-1), min(256,nfit-1), min(511,nfit-1), min(512,nfit-1), min(768,nfit-1), nfit-1)])

# ---- backward kernel
out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
dO = torch.randn_like(out)
for _ in range(2):
    gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO)
sb = torch.zeros(nfit * 16, dtype=torch.int64, device=dev)
L.dfgnn_debug_set_bwd_stamps.argtypes = [ctypes.c_void_p]
assert L.dfgnn_debug_set_bwd_stamps(sb.data_ptr()) == 0
torch.cuda.synchronize()
e0.record(); gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO); e1.record()
torch.cuda.synchronize()
tb = sb.cpu().numpy().reshape(nfit, 16)[:, :12].astype(np.float64)
db = np.diff(tb, axis=1)
bn = ["stage idx+V", "pass1 dP/dS", "barrier", "load K", "pass2 dQ", "barrier", "load dO", "pass3 dV", "barrier", "load Q", "pass4 dK"]
print("BWD kernel ms", e0.elapsed_time(e1))
for k, nm in enumerate(bn):
    print(f"{nm:14s} mean {db[:,k].mean():9.0f}  p50 {np.median(db[:,k]):9.0f}  max {db[:,k].max():9.0f}")
totb = tb[:, 11] - tb[:, 0]
print("bwd total per WG mean", totb.mean(), "max", totb.max(), " sum/256:", totb.sum() / 256)
