import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden():
    out = {}
    for fn in sorted(os.listdir(GOLDEN_DIR)):
        if fn.endswith(".npz"):
            out[fn[:-4]] = dict(np.load(os.path.join(GOLDEN_DIR, fn), allow_pickle=False))
    assert out, "no golden fixtures found"
    return out


@pytest.fixture(scope="session")
def golden_gat_train(golden):
    """GAT training-pair expectations for the fixtures of `golden` (tests/golden/make_golden_gat_train.py)."""
    d = os.path.join(GOLDEN_DIR, "gat_train")
    out = {fn[:-4]: dict(np.load(os.path.join(d, fn), allow_pickle=False)) for fn in sorted(os.listdir(d))
           if fn.endswith(".npz")}
    assert sorted(out) == sorted(golden), "gat_train fixtures out of step with tests/golden/*.npz"
    return out


# ---- tiny graph builders shared by CPU and GPU tests ------------------------------------------
def random_graph(rng, m, avg_deg, empty_frac=0.0, dup_frac=0.0, max_deg=None):
    """Random CSR graph (int32) with optional empty rows / duplicate edges / one heavy row."""
    deg = rng.poisson(avg_deg, m)
    if empty_frac > 0:
        deg[rng.random(m) < empty_frac] = 0
    if max_deg is not None and m > 0:
        deg[rng.integers(m)] = max_deg
    indptr = np.zeros(m + 1, dtype=np.int32)
    indptr[1:] = np.cumsum(deg)
    nnz = int(indptr[-1])
    indices = rng.integers(0, max(m, 1), nnz).astype(np.int32)
    if dup_frac > 0 and nnz > 1:
        k = int(nnz * dup_frac)
        pos = rng.integers(1, nnz, k)
        same_row = np.searchsorted(indptr, pos, side="right") == np.searchsorted(indptr, pos - 1, side="right")
        indices[pos[same_row]] = indices[pos[same_row] - 1]
    rows = np.repeat(np.arange(m, dtype=np.int32), deg)
    return indptr, indices, rows


def csc_of(indptr, indices, rows, m):
    order = np.argsort(indices, kind="stable")
    col_ptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(col_ptr, indices.astype(np.int64) + 1, 1)
    return np.cumsum(col_ptr).astype(np.int32), rows[order].astype(np.int32), order.astype(np.int32)
