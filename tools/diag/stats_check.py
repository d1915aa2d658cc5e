#!/usr/bin/env python3
"""Quick correctness + timing check of the statistics-saving GT training pair against the oracle and the attn_edge pair.
usage: python tools/diag/stats_check.py [--time]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt
import oracle
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import Graph, batch, synthetic as S
DEV = "cuda:0"


def geometry_batch(seed):
    rng = np.random.default_rng(seed)
    def er(n, p, drop=()):
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(len(iu)) < p
        keep &= ~np.isin(iu, drop) & ~np.isin(ju, drop)
        return np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]])
    graphs = []
    for n, p in ((9, 0.9), (17, 0.6), (64, 0.5), (128, 0.35), (129, 0.3), (145, 0.4), (160, 0.3), (161, 0.3), (200, 0.25), (255, 0.2)):
        graphs.append(Graph(*er(n, p), n))
    graphs.append(Graph(*er(70, 0.5, drop=(0, 33, 69)), 70))
    keep = rng.random((80, 140)) < 0.3
    ds_, dd_ = np.nonzero(keep)
    graphs.append(Graph(ds_.astype(np.int64), dd_.astype(np.int64), 140))   # directed: mask != maskT
    keep = rng.random((150, 150)) < 0.2
    ds_, dd_ = np.nonzero(keep)
    graphs.append(Graph(ds_.astype(np.int64), dd_.astype(np.int64), 150))
    return batch(graphs).to(DEV)


def check(h, f):
    g = geometry_batch(17 + f)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, h, f, seed=5, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    assert gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q), "stats pair does not apply"
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    dQ, dK, dV = gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)
    torch.cuda.synchronize()
    n_ = lambda t: t.cpu().numpy()
    want = oracle.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V))
    wq, wk, wv = oracle.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    res = {}
    for k, a, b in (("out", out, want), ("dQ", dQ, wq), ("dK", dK, wk), ("dV", dV, wv)):
        res[k] = float(np.abs(n_(a).astype(np.float64) - b).max())
    print(f"h={h} f={f}: " + " ".join(f"{k} {v:.2e}" for k, v in res.items()), flush=True)
    return max(res.values())


def timing(h, f, bs=1024):
    g = S.pattern_like(batch_size=bs, seed=1).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, h, f, seed=5, device=DEV)
    dO = torch.randn(m, h, f, device=DEV)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    def ev(fn, reps=20):
        for _ in range(5): fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
    out, attn = gt.gt_hyper_forward(*args)
    o2, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    r = dict(h=h, f=f, bs=bs,
             fwd_attn=ev(lambda: gt.gt_hyper_forward(*args)), bwd_attn=ev(lambda: gt.gt_backward(*args, attn, dO)),
             fwd_stats=ev(lambda: gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)),
             bwd_stats=ev(lambda: gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)),
             fwd_inf=ev(lambda: gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)))
    print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)


if __name__ == "__main__":
    worst = 0.0
    for h, f in ((1, 128), (1, 64), (2, 64), (4, 32), (8, 16), (3, 32), (2, 8), (2, 128)):
        worst = max(worst, check(h, f))
    print("worst", worst)
    if "--time" in sys.argv:
        for h, f in ((1, 128), (2, 64), (4, 32), (8, 16)):
            timing(h, f)
        timing(1, 128, bs=128)
