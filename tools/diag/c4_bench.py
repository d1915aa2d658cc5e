#!/usr/bin/env python3
"""C4 (reddit-like GAT 'tiling', f = 128): single-kernel form vs column chunks of several sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import fused_gatconv as gat
from DFGNN.layers.util import preprocess_CSR
from DFGNN.utils import synthetic as S
dev = "cuda:0"
t0 = time.perf_counter()
g = S.reddit_like().to(dev)
row_ptr, col_ind, val, _ = preprocess_CSR(g)
m, nnz = g.num_nodes(), g.num_edges()
print(f"graph built in {time.perf_counter()-t0:.1f}s: m={m} nnz={nnz}", flush=True)
ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=dev)


def ev(fn, reps=5):
    for _ in range(2):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in evs]))


gat.TILING_CHUNK_ROWS = 0
gat.TILING_CHUNK_MIN_TABLE = 0
ref = gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)
print(f"single kernel: {ev(lambda: gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)):.3f} ms", flush=True)
for cr in [int(a) for a in sys.argv[1:]] or [8192, 4096, 16384]:
    gat.TILING_CHUNK_ROWS = cr
    t0 = time.perf_counter()
    out = gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    ms = ev(lambda: gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X))
    print(f"chunk_rows={cr}: {ms:.3f} ms  (first call incl. chunk build {t_first*1e3:.0f} ms)  max |diff| vs single kernel "
          f"{(out - ref).abs().max().item():.2e}", flush=True)
    gat._chunk_cache.d.clear()
