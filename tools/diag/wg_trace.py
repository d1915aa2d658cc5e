#!/usr/bin/env python3
"""Diagnostic (needs the -DDFGNN_STAMPS build, tools/diag/build_variant.sh stamps -DDFGNN_STAMPS): which CU ran which
workgroup of the matrix-core kernels and when -- how much of the kernel's span a CU has a workgroup resident, the gaps
between consecutive workgroups on a CU, the ragged tail.  Only entry / exit times are taken (no phase stamps are armed),
so the kernels run close to their normal speed.
usage: python3 tools/diag/wg_trace.py [batch_size]"""
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
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = S.pattern_like(batch_size=bs, seed=1).to(dev)
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = S.gt_features(m, 1, 128, seed=100, device=dev)
L = dfgnn_native.lib()
L.dfgnn_debug_set_wg_trace.argtypes = [ctypes.c_void_p]
args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
for _ in range(3):
    out, attn = gt.gt_hyper_forward(*args)
dO = torch.randn_like(out)
for _ in range(3):
    gt.gt_backward(*args, attn, dO)
nd = row_ptr._dfgnn_plans[128].num_dense
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)


def trace(label, fn):
    tr = torch.zeros(nd * 8, dtype=torch.int64, device=dev)
    assert L.dfgnn_debug_set_wg_trace(tr.data_ptr()) == 0
    torch.cuda.synchronize()
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    assert L.dfgnn_debug_set_wg_trace(None) == 0
    t = tr.cpu().numpy().reshape(nd, 8)
    hw = t[:, 2]
    ghz = np.median((t[:, 1] - t[:, 0]).astype(np.float64) / np.maximum(1, (t[:, 4] - t[:, 3]).astype(np.float64) * 10.0))
    # the timeline is on the 100 MHz clock (one tick = 10 ns, the same on every XCC), printed in ns
    t_in, t_out = t[:, 3].astype(np.float64) * 10.0, t[:, 4].astype(np.float64) * 10.0
    xcc = (hw >> 32) & 0xF
    cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ms = e0.elapsed_time(e1)
    print(f"{label}: {ms * 1e3:.1f} us by events; {len(np.unique(cuid))} CUs used, {len(np.unique(xcc))} XCCs; "
          f"s_memtime runs at ~{ghz:.2f} GHz (residency on both clocks)")
    k0 = t_in.min(); k1 = t_out.max()
    span = k1 - k0
    busy = (t_out - t_in).sum()
    ncu = len(np.unique(cuid))
    print(f"  kernel span (first entry .. last exit) {span:.0f} ns; resident time summed {busy:.0f} = "
          f"{busy / (ncu * span) * 100:.1f} % of {ncu} CUs x span")
    head, gaps, tail, per = [], [], [], []
    for c in np.unique(cuid):
        sel = np.where(cuid == c)[0]
        o = sel[np.argsort(t_in[sel])]
        head.append(t_in[o[0]] - k0)
        tail.append(k1 - t_out[o[-1]])
        per.append(len(o))
        for a, b in zip(o[:-1], o[1:]):
            gaps.append(t_in[b] - t_out[a])
    gaps = np.array(gaps); head = np.array(head); tail = np.array(tail)
    print(f"  workgroups per CU: min {min(per)} max {max(per)}; head (first entry after kernel start) mean {head.mean():.0f} max {head.max():.0f}; "
          f"gap between consecutive workgroups on a CU mean {gaps.mean():.0f} p50 {np.median(gaps):.0f} p90 {np.percentile(gaps, 90):.0f} "
          f"(overlaps count negative: min {gaps.min():.0f}); tail (idle before the kernel's last exit) mean {tail.mean():.0f} max {tail.max():.0f}")
    print(f"  per CU of the span: head {head.mean() / span * 100:.1f} %, gaps {gaps.sum() / ncu / span * 100:.1f} %, tail {tail.mean() / span * 100:.1f} %; "
          f"workgroup residency mean {np.mean(t_out - t_in):.0f} ns")


trace("forward", lambda: gt.gt_hyper_forward(*args))
trace("backward", lambda: gt.gt_backward(*args, attn, dO))
