#!/usr/bin/env python3
"""A/B of library builds on the statistics-saving pair: kernel times on the PATTERN-like mix and on equal-graph classes.
usage: ab_stats.py lib1.so lib2.so ... [--heads H] [--classes 107,128]   (each library in a child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--child" not in sys.argv:
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    rest = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for lib in libs:
        env = dict(os.environ, DFGNN_LIB=lib, DFGNN_BINDING="ctypes")
        subprocess.run([sys.executable, __file__, "--child"] + rest, env=env, check=False)
    sys.exit(0)
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gtconv as gt
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.utils import synthetic as S
dev = "cuda:0"
argv = sys.argv[1:]
H = 1
classes = []
if "--heads" in argv:
    H = int(argv[argv.index("--heads") + 1])
if "--classes" in argv:
    classes = [int(x) for x in argv[argv.index("--classes") + 1].split(",")]
f = 128 // H


def ev(fn, reps=30):
    for _ in range(5):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3


def one(g, label):
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, H, f, seed=100, device=dev)
    dO = torch.randn_like(Q)
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    sf = ev(lambda: gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V))
    sb = ev(lambda: gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO))
    print(f"{os.environ.get('DFGNN_LIB'):28s} h={H} {label:8s}: stats fwd {sf:6.1f} bwd {sb:6.1f} us", flush=True)


one(S.pattern_like(batch_size=1024, seed=1).to(dev), "mix")
for n in classes:
    one(S.pattern_like(batch_size=1024, seed=1, mean_nodes=float(n), std_nodes=0.0, lo=n, hi=n, mean_deg=0.43 * (n - 1)).to(dev), f"n={n}")
