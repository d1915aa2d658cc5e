#!/usr/bin/env python3
"""Per-workgroup phase times of the statistics-saving pair's kernels (needs tools/diag/build_stats_stamps.sh).
usage: stamps_stats.py [--heads H] [n]     (n: batch of equal graphs of n nodes; default: the PATTERN-like mix)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
os.environ["DFGNN_BINDING"] = "ctypes"
os.environ["DFGNN_LIB"] = "libdfgnn_sstamps.so"
os.environ["DFGNN_LEAN"] = "0"
import dfgnn_native
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
argv = sys.argv[1:]
H = 1
if "--heads" in argv:
    k = argv.index("--heads"); H = int(argv[k + 1]); del argv[k:k + 2]
FH = 128 // H
bs = 1024
if argv:
    n = int(argv[0])
    g = S.pattern_like(batch_size=bs, seed=1, mean_nodes=float(n), std_nodes=0.0, lo=n, hi=n, mean_deg=0.43 * (n - 1)).to(dev)
else:
    g = S.pattern_like(batch_size=bs, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, H, FH, seed=100, device=dev)
dO = torch.randn_like(Q)
L = dfgnn_native.lib()
L.dfgnn_debug_set_dense_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)
plan = row_ptr._dfgnn_plans[FH]
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)


def run(fn, nwg, order, names, rows_sel=None):
    st = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
    assert L.dfgnn_debug_set_dense_stamps(st.data_ptr()) == 0
    torch.cuda.synchronize()
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    assert L.dfgnn_debug_set_dense_stamps(None) == 0
    s = st.cpu().numpy().reshape(nwg, 16)
    if rows_sel is not None:
        s = s[rows_sel]
    s = s[(s[:, order] != 0).all(axis=1)]
    t = s[:, order].astype(np.float64)
    d = np.diff(t, axis=1)
    tot = t[:, -1] - t[:, 0]
    print(f"  {e0.elapsed_time(e1)*1e3:.1f} us; {len(s)} workgroups; cycles per WG mean {tot.mean():.0f} p50 {np.median(tot):.0f} max {tot.max():.0f}")
    print("    " + "  ".join(f"{nm} {np.median(d[:, k]):.0f}" for k, nm in enumerate(names)))
    print("    sum of WG cycles / 256 CUs:", round(tot.sum() / 256), " kernel span:", t[:, -1].max() - t[:, 0].min())


nd = plan.num_dense
if H == 1:
    print("FWD stats (512-thread body)")
    run(lambda: gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V), nd, [0, 1, 2, 3, 4, 5],
        ["prologue+K img", "S", "softmax", "V img", "PV+store"])
    print("BWD stats (rc2 body: ranges of <= 128 nodes only are stamped)")
    run(lambda: gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO), nd, [0, 1, 2, 3, 4, 5, 6, 7],
        ["[dO|V] img", "dP", "[Q|K] img", "S,P,dS,dQ", "dS->tile,dK", "dO img,P->tile", "dV"])
else:
    print("BWD stats multi-head (heads2 body)")
    sel = np.arange(nd) * H   # the workgroups of head 0 (stamp row = blockIdx.x * gridDim.y + blockIdx.y)
    run(lambda: gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO), nd * H, [0, 1, 2, 3, 4, 5],
        ["prologue+K,V img", "row pass", "Q,dO img", "col pass", "other groups"], rows_sel=sel)
