#!/usr/bin/env python3
"""Preprocessing cost per batch on the C3 workload (PATTERN-like, bs = 1024): COO -> CSR / rows / CSC through the
native GPU path (dfgnn_preprocess_hyper) vs the torch restatement of the dgl.sparse calls on the same device, and the
block-plan build.  The reference counts this inside a training epoch (train_batch_graph_timing.py:115-143).
Wall time per call around a device synchronisation (the calls are host-driven sequences of launches), 3 dry + 10
timed.  Prints one JSON line.   usage: python3 tools/bench_prep.py [batch_size]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import dfgnn_preprocess  # noqa: E402
from _binding_util import build_plan  # noqa: E402
from DFGNN.utils import sparse as dglsp  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"
g = S.pattern_like(batch_size=bs, seed=1).to(dev)
src, dst = g.edges()
n, nnz = g.num_nodes(), g.num_edges()


def torch_path():
    A = dglsp.spmatrix(torch.stack((src, dst)), shape=(n, n))
    rows = torch.sort(A.row.int()).values
    row_ptr, col_ind, val_idx = A.csr()
    row_ptr, col_ind, val = row_ptr.int(), col_ind.int(), A.val[val_idx]
    col_ptr, row_ind, vi = dglsp.from_csr(indptr=row_ptr, indices=col_ind, val=val, shape=A.shape).csc()
    return rows, row_ptr, col_ind, val, col_ptr.int(), row_ind.int(), vi.int()


def native_path():
    return dfgnn_preprocess.coo_to_hyper(src, dst, n, csc=True)


def timed(fn, dry=3, it=10):
    for _ in range(dry):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


row_ptr, col_ind = native_path()[:2]
out = dict(workload=f"PATTERN-like bs={bs}", nodes=n, edges=nnz, torch_ms=timed(torch_path), native_ms=timed(native_path),
           plan_ms=timed(lambda: build_plan(row_ptr, col_ind, 128)))
out["speedup"] = out["torch_ms"] / out["native_ms"]
print(json.dumps(out))
