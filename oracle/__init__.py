"""CPU oracle for the fused attention-GNN convolution (SDDMM -> edge-softmax -> SpMM).

TEST INFRASTRUCTURE ONLY -- see the header of ``oracle/oracle.c``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product path under ``df-gnn_amd/`` never does.

PARITY UNPINNED: the reference ships no golden vectors for this path and neither its CUDA
kernels nor its dgl.sparse ground truth can run here.  The C restatement below
is cross-checked against ``oracle/torch_ref.py`` (an independently written edge-list/autograd
restatement of ``forward_dglsp``) and hand-derived known-answer cases in ``tests/``.

Functions take/return numpy arrays; features are ``[m, h, f]`` float32, CSR is int32.
Outputs are float64 (``acc="f64"``, the checker) or float32 (``acc="f32"``, the timed baseline).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile ``liboracle.so`` with gcc (recipe: oracle/Makefile)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def num_threads():
    return int(_lib().oracle_num_threads_f64())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _acc(acc):
    if acc == "f64":
        return "_f64", np.float64
    if acc == "f32":
        return "_f32", np.float32
    raise ValueError(acc)


def gt_forward(indptr, indices, val, Q, K, V, want_attn=False, acc="f64"):
    """Oracle of ``gt_hyper_inference`` / ``gt_hyper_forward`` (and every GT variant: they all
    compute the same function).  Returns ``out[m,h,f]`` (and ``attn_edge[h,nnz]``)."""
    sfx, dt = _acc(acc)
    indptr, indices, val = _i32(indptr), _i32(indices), _f32(val)
    Q, K, V = _f32(Q), _f32(K), _f32(V)
    m, nnz = indptr.shape[0] - 1, indices.shape[0]
    _, h, f = Q.shape
    out = np.empty((m, h, f), dtype=dt)
    attn = np.empty((h, nnz), dtype=dt) if want_attn else None
    getattr(_lib(), "oracle_gt_forward" + sfx)(
        ctypes.c_int(m), ctypes.c_int(nnz), ctypes.c_int(h), ctypes.c_int(f),
        _p(indptr), _p(indices), _p(val), _p(Q), _p(K), _p(V), _p(out), _p(attn))
    return (out, attn) if want_attn else out


def gt_backward(indptr, indices, val, Q, K, V, dO, acc="f64"):
    """Oracle of ``gt_backward``: returns ``(dQ, dK, dV)``, each ``[m,h,f]``."""
    sfx, dt = _acc(acc)
    indptr, indices, val = _i32(indptr), _i32(indices), _f32(val)
    Q, K, V, dO = _f32(Q), _f32(K), _f32(V), _f32(dO)
    m, nnz = indptr.shape[0] - 1, indices.shape[0]
    _, h, f = Q.shape
    dQ, dK, dV = (np.empty((m, h, f), dtype=dt) for _ in range(3))
    getattr(_lib(), "oracle_gt_backward" + sfx)(
        ctypes.c_int(m), ctypes.c_int(nnz), ctypes.c_int(h), ctypes.c_int(f),
        _p(indptr), _p(indices), _p(val), _p(Q), _p(K), _p(V), _p(dO), _p(dQ), _p(dK), _p(dV))
    return dQ, dK, dV


def gat_forward(indptr, indices, attn_row, attn_col, negative_slope, X, want_attn=False, acc="f64"):
    """Oracle of ``gat_inference_{hyper,softmax,softmax_gm,tiling}``: returns ``out[m,h,f]``."""
    sfx, dt = _acc(acc)
    indptr, indices = _i32(indptr), _i32(indices)
    attn_row, attn_col, X = _f32(attn_row), _f32(attn_col), _f32(X)
    m, nnz = indptr.shape[0] - 1, indices.shape[0]
    _, h, f = X.shape
    out = np.empty((m, h, f), dtype=dt)
    attn = np.empty((h, nnz), dtype=dt) if want_attn else None
    getattr(_lib(), "oracle_gat_forward" + sfx)(
        ctypes.c_int(m), ctypes.c_int(nnz), ctypes.c_int(h), ctypes.c_int(f),
        _p(indptr), _p(indices), _p(attn_row), _p(attn_col), ctypes.c_float(negative_slope),
        _p(X), _p(out), _p(attn))
    return (out, attn) if want_attn else out


def gat_train_forward(indptr, indices, attn_row, attn_col, negative_slope, X, edge_mask=None, attn_drop=0.0,
                      acc="f64"):
    """Oracle of ``gat_forward`` (FusedGATFunction.forward, DFGNN/operators/fused_gatconv.py:95-127): returns
    ``(out[m,h,f], edge_max[m,h], edge_sum[m,h])``.  ``edge_mask`` is the uniform-random tensor ``[nnz, h]`` the
    reference draws with cuRAND (an input here, so results are reproducible); ``None`` = no dropout."""
    sfx, dt = _acc(acc)
    indptr, indices = _i32(indptr), _i32(indices)
    attn_row, attn_col, X = _f32(attn_row), _f32(attn_col), _f32(X)
    mask = _f32(edge_mask) if edge_mask is not None else None
    m, nnz = indptr.shape[0] - 1, indices.shape[0]
    _, h, f = X.shape
    out = np.empty((m, h, f), dtype=dt)
    emax = np.empty((m, h), dtype=dt)
    esum = np.empty((m, h), dtype=dt)
    getattr(_lib(), "oracle_gat_train_forward" + sfx)(
        ctypes.c_int(m), ctypes.c_int(nnz), ctypes.c_int(h), ctypes.c_int(f),
        _p(indptr), _p(indices), _p(attn_row), _p(attn_col), ctypes.c_float(negative_slope),
        _p(X), _p(mask), ctypes.c_float(attn_drop), _p(out), _p(emax), _p(esum))
    return out, emax, esum


def gat_backward(indptr, indices, attn_row, attn_col, negative_slope, X, dO, edge_mask=None, attn_drop=0.0,
                 acc="f64"):
    """Oracle of ``gat_backward``: returns ``(grad_feat[m,h,f], grad_attn_row[m,h], grad_attn_col[m,h])``
    (FusedGATFunction.backward, DFGNN/operators/fused_gatconv.py:129-176)."""
    sfx, dt = _acc(acc)
    indptr, indices = _i32(indptr), _i32(indices)
    attn_row, attn_col, X, dO = _f32(attn_row), _f32(attn_col), _f32(X), _f32(dO)
    mask = _f32(edge_mask) if edge_mask is not None else None
    m, nnz = indptr.shape[0] - 1, indices.shape[0]
    _, h, f = X.shape
    gf = np.empty((m, h, f), dtype=dt)
    gr = np.empty((m, h), dtype=dt)
    gc = np.empty((m, h), dtype=dt)
    getattr(_lib(), "oracle_gat_backward" + sfx)(
        ctypes.c_int(m), ctypes.c_int(nnz), ctypes.c_int(h), ctypes.c_int(f),
        _p(indptr), _p(indices), _p(attn_row), _p(attn_col), ctypes.c_float(negative_slope),
        _p(X), _p(mask), ctypes.c_float(attn_drop), _p(dO), _p(gf), _p(gr), _p(gc))
    return gf, gr, gc


# ---- preprocessing oracle (layers/util.py:82-142 of the reference, restated with numpy) -------
def coo_to_hyper(src, dst, num_nodes):
    """(src, dst) edge list -> dict with the reference's preprocess_Hyper_fw_bw arrays.

    Follows DFGNN/layers/util.py:52-57,116-142: row = src, col = dst; CSR by stable row sort
    (ties keep edge-list order); ``rows`` = sorted COO row ids; CSC of the CSR matrix with
    ``val_idx`` = CSR slot of each CSC entry (stable in CSR order)."""
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    order = np.argsort(src, kind="stable")
    rows = src[order].astype(np.int32)
    col_ind = dst[order].astype(np.int32)
    row_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr).astype(np.int32)
    corder = np.argsort(col_ind, kind="stable")
    val_idx = corder.astype(np.int32)
    row_ind = rows[corder].astype(np.int32)
    col_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(col_ptr, dst + 1, 1)
    col_ptr = np.cumsum(col_ptr).astype(np.int32)
    return dict(rows=rows, row_ptr=row_ptr, col_ind=col_ind, val=np.ones(len(src), np.float32),
                col_ptr=col_ptr, row_ind=row_ind, val_idx=val_idx, edge_order=order)
