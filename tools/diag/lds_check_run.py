#!/usr/bin/env python3
"""Run the matrix-core kernels of the bounds-checking build (csrc `make ldscheck`, df-gnn_amd/libdfgnn_ldscheck.so) over every
range class and head layout and print what dfgnn_debug_lds_report() counted, as one JSON line.
usage: DFGNN_LIB=libdfgnn_ldscheck.so python3 tools/diag/lds_check_run.py"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
assert os.environ.get("DFGNN_LIB") == "libdfgnn_ldscheck.so", "run with DFGNN_LIB=libdfgnn_ldscheck.so"
import torch  # noqa: E402

import dfgnn_native  # noqa: E402
import fused_gatconv as gat  # noqa: E402
import fused_gtconv as gt  # noqa: E402
from DFGNN.layers import preprocess_Hyper_fw_bw  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402

dev = "cuda:0"
L = dfgnn_native.lib()
assert dfgnn_native.ext() is None           # the extension binds libdfgnn.so: this run must stay on the library named above
L.dfgnn_debug_lds_report.argtypes = [ctypes.POINTER(ctypes.c_uint)]
L.dfgnn_debug_lds_selftest.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]


def report():
    r = (ctypes.c_uint * 4)()
    units = L.dfgnn_debug_lds_report(r)
    return {"units": units, "violations": r[0], "line": r[1], "end": r[2], "limit": r[3]}


out = {"build_id": dfgnn_native.build_id(), "source_hash": dfgnn_native.source_hash(), "start": report(), "cases": []}
buf = torch.zeros(4, device=dev)
assert L.dfgnn_debug_lds_selftest(4096, buf.data_ptr(), None) == 0
torch.cuda.synchronize()
out["selftest"] = dict(report(), readback=float(buf[0]))     # one counted store just past 4096 bytes, limit 4096
launches = 0
for n in (40, 107, 128, 140, 160, 175, 250):                  # every range class of the plan's dense kernels
    g = S.pattern_like(batch_size=12, seed=n, mean_nodes=float(n), std_nodes=0.0, lo=n, hi=n, mean_deg=0.43 * (n - 1)).to(dev)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    for h, f in ((1, 128), (1, 64), (1, 32), (2, 64), (4, 32), (8, 16), (2, 128)):
        Q, K, V = S.gt_features(m, h, f, seed=n + h, device=dev)
        dO = torch.randn_like(Q)
        args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        o, attn = gt.gt_hyper_forward(*args)                                  # attn_edge pair
        gt.gt_backward(*args, attn, dO)
        gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)
        launches += 3
        if gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q) is not None:    # statistics pair
            o2, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
            gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)
            launches += 2
        if gt.gt_ranked_pair_applies(row_ptr, col_ind, val, Q) is not None:   # attn_edge pair in rank order (one head)
            o3, attn_r = gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V)
            gt.gt_backward_ranked(row_ptr, col_ind, Q, K, V, attn_r, dO)
            launches += 2
        if h == 1:                                                            # GAT training pair, with and without dropout
            ar, ac, X = S.gat_features(m, 1, f, seed=n, device=dev)
            for drop in (0.0, 0.3):
                res = gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, drop)
                gat.gat_backward(0.2, drop, row_ptr, col_ind, col_ptr, row_ind, val_idx, res[1], res[2], res[3], X, ar, ac, dO)
                launches += 2
    torch.cuda.synchronize()
    out["cases"].append(dict(report(), nodes_per_graph=n))
out["launches"] = launches
print(json.dumps(out))
