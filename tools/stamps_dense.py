#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the matrix-core kernels (gt_dense.hip; needs the -DDFGNN_STAMPS build)."""
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
H = int(os.environ.get("HEADS", "1"))   # HEADS=8: dim 128 as 8 heads of 16 (backward: one workgroup per (range, head))
FH = 128 // H
g = S.pattern_like(batch_size=bs, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, H, FH, seed=100, device=dev)
L = dfgnn_native.lib()
L.dfgnn_debug_set_dense_stamps.argtypes = [ctypes.c_void_p]
args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
for _ in range(3):
    out, attn = gt.gt_hyper_forward(*args)
plan = row_ptr._dfgnn_plans[FH]
nd = plan.num_dense * H   # (stamp rows: one per workgroup)
print("plan", plan.meta)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)


def run(fn, names):
    st = torch.zeros(nd * 16, dtype=torch.int64, device=dev)
    assert L.dfgnn_debug_set_dense_stamps(st.data_ptr()) == 0
    torch.cuda.synchronize()
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    assert L.dfgnn_debug_set_dense_stamps(None) == 0
    s = st.cpu().numpy().reshape(nd, 16)
    n = s[:, 15] >> 32; ne = s[:, 15] & 0xffffffff
    t = s[:, :len(names) + 1].astype(np.float64)
    d = np.diff(t, axis=1)
    print("  ms", e0.elapsed_time(e1))
    for sel, lab in ((n <= 128, "n<=128"), (n > 128, "n>128")):
        if sel.sum() == 0:
            continue
        tot = t[sel, len(names)] - t[sel, 0]
        print(f"  {lab}: {sel.sum()} ranges, nodes mean {n[sel].mean():.0f}, edges mean {ne[sel].mean():.0f}; "
              f"cycles per WG mean {tot.mean():.0f} p50 {np.median(tot):.0f} max {tot.max():.0f}")
        print("    " + "  ".join(f"{nm} {np.median(d[sel, k]):.0f}" for k, nm in enumerate(names)))
    tot = t[:, len(names)] - t[:, 0]
    print("  sum of WG cycles / 256 CUs:", tot.sum() / 256, " kernel span:", t[:, len(names)].max() - t[:, 0].min())


print("FWD (training)")
run(lambda: gt.gt_hyper_forward(*args), ["prologue+map", "S", "softmax+attn", "V stage", "PV+store"])
dO = torch.randn_like(out)
for _ in range(2):
    gt.gt_backward(*args, attn, dO)
print("BWD (stamps of the last row block)")
# stamp order in the kernel (last row block): 0 start, 9 tile loaded, 3 P tile ready, 4 dV done, 1 dP done, 2 dS done,
# 5 dS tile + K image ready, 6 dQ done, 7 Q image ready, 8 dK done
def run_bwd():
    st = torch.zeros(nd * 16, dtype=torch.int64, device=dev)
    assert L.dfgnn_debug_set_dense_stamps(st.data_ptr()) == 0
    torch.cuda.synchronize()
    e0.record(); gt.gt_backward(*args, attn, dO); e1.record(); torch.cuda.synchronize()
    assert L.dfgnn_debug_set_dense_stamps(None) == 0
    s = st.cpu().numpy().reshape(nd, 16)
    n = s[:, 15] >> 32; ne = s[:, 15] & 0xffffffff
    order = [0, 9, 3, 4, 1, 2, 5, 6, 7, 8]
    names = ["tile+dO img", "P->tile", "dV", "V img,dP", "t,dS", "dS->tile,K img", "dQ", "Q img", "dK"]
    t = s[:, order].astype(np.float64)
    d = np.diff(t, axis=1)
    print("  ms", e0.elapsed_time(e1))
    for sel, lab in ((n <= 128, "n<=128"), (n > 128, "n>128 (phases of the LAST row block; the first column = everything before)")):
        if sel.sum() == 0:
            continue
        tot = s[sel, 8].astype(np.float64) - s[sel, 0]
        print(f"  {lab}: {sel.sum()} ranges, nodes mean {n[sel].mean():.0f}, edges mean {ne[sel].mean():.0f}; "
              f"cycles per WG mean {tot.mean():.0f} p50 {np.median(tot):.0f} max {tot.max():.0f}")
        print("    " + "  ".join(f"{nm} {np.median(d[sel, k]):.0f}" for k, nm in enumerate(names)))
    tot = s[:, 8].astype(np.float64) - s[:, 0]
    print("  sum of WG cycles / 256 CUs:", tot.sum() / 256)


print("BWD")
run_bwd()
