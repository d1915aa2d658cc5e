#!/usr/bin/env python3
"""Per-workgroup phase times of the rank-ordered forward (needs a stamp build of gt_dense_stats_w.hip:
tools/diag/build_w_variant.sh wst -DDFGNN_STAMPS -DDFGNN_STAMPS_TU [-DDFGNN_RANKED_WRITE=false]).
usage: DFGNN_LIB=libdfgnn_wst.so stamps_ranked.py [n]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
os.environ["DFGNN_BINDING"] = "ctypes"
import dfgnn_native
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
argv = sys.argv[1:]
if argv:
    n = int(argv[0])
    g = S.pattern_like(batch_size=1024, seed=1, mean_nodes=float(n), std_nodes=0.0, lo=n, hi=n, mean_deg=0.43 * (n - 1)).to(dev)
else:
    g = S.pattern_like(batch_size=1024, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
Q, K, V = S.gt_features(g.num_nodes(), 1, 128, seed=100, device=dev)
L = dfgnn_native.lib()
L.dfgnn_debug_set_dense_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V)
nwg = row_ptr._dfgnn_plans[128].num_dense
st = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
assert L.dfgnn_debug_set_dense_stamps(st.data_ptr()) == 0
torch.cuda.synchronize()
e0.record(); gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V); e1.record(); torch.cuda.synchronize()
assert L.dfgnn_debug_set_dense_stamps(None) == 0
s = st.cpu().numpy().reshape(nwg, 16)
order = [0, 1, 2, 3, 4, 5, 6]
s = s[(s[:, order] != 0).all(axis=1)]
t = s[:, order].astype(np.float64)
d = np.diff(t, axis=1)
tot = t[:, -1] - t[:, 0]
names = ["prologue+K img", "S", "softmax(+attn out)", "V img", "P.V+store", "tail"]
print(f"{e0.elapsed_time(e1)*1e3:.1f} us; {len(s)} workgroups; cycles per WG mean {tot.mean():.0f} p50 {np.median(tot):.0f}")
print("  " + "  ".join(f"{nm} {np.median(d[:, k]):.0f}" for k, nm in enumerate(names)))
