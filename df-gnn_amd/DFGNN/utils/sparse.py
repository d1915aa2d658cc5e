"""The few `dgl.sparse` calls the reference's *non-fused* branches make, restated in torch.

The reference's `fuse=False` paths run dgl.sparse on the GPU as the comparison baseline
(DFGNN/layers/GT/gtconv_layer.py:29-33, DFGNN/layers/GAT/gatconv_layer.py:30-38).  dgl is not
available on ROCm boxes here, so the layers' baseline branch uses these torch equivalents
(`import DFGNN.utils.sparse as dglsp`).  They run on whatever device the tensors live on.
This is the baseline the fused kernels are *compared with* in the harness; it is not the oracle
(oracle/ is test infrastructure and is never imported from here).
"""
import torch


class SparseMatrix:
    """COO sparse matrix with per-nnz values of shape [nnz] or [nnz, heads]."""

    def __init__(self, row, col, val, shape):
        self.row, self.col, self.val, self.shape = row, col, val, tuple(shape)

    @property
    def nnz(self):
        return int(self.row.numel())

    @property
    def device(self):
        return self.row.device

    def softmax(self):
        """Row-wise softmax over the non-zeros (dgl.sparse.SparseMatrix.softmax)."""
        v = self.val
        squeeze = v.dim() == 1
        if squeeze:
            v = v.unsqueeze(1)
        m = self.shape[0]
        idx = self.row.unsqueeze(1).expand(-1, v.shape[1])
        mx = torch.full((m, v.shape[1]), float("-inf"), dtype=v.dtype, device=v.device)
        mx = mx.scatter_reduce(0, idx, v, reduce="amax", include_self=True)
        ex = torch.exp(v - mx[self.row])
        den = torch.zeros((m, v.shape[1]), dtype=v.dtype, device=v.device).index_add_(0, self.row, ex)
        out = ex / den[self.row]
        return SparseMatrix(self.row, self.col, out.squeeze(1) if squeeze else out, self.shape)

    def csr(self):
        """(indptr, indices, value_indices): stable sort by row, like dgl's COO->CSR."""
        order = torch.argsort(self.row, stable=True)
        counts = torch.bincount(self.row, minlength=self.shape[0])
        indptr = torch.zeros(self.shape[0] + 1, dtype=torch.int64, device=self.device)
        indptr[1:] = torch.cumsum(counts, 0)
        return indptr, self.col[order], order

    def csc(self):
        order = torch.argsort(self.col, stable=True)
        counts = torch.bincount(self.col, minlength=self.shape[1])
        indptr = torch.zeros(self.shape[1] + 1, dtype=torch.int64, device=self.device)
        indptr[1:] = torch.cumsum(counts, 0)
        return indptr, self.row[order], order


def spmatrix(indices, val=None, shape=None):
    """dgl.sparse.spmatrix: `indices` is [2, nnz] (row; col); values default to ones."""
    row, col = indices[0], indices[1]
    if val is None:
        val = torch.ones(row.numel(), dtype=torch.float32, device=row.device)
    if shape is None:
        n = int(max(row.max().item(), col.max().item())) + 1 if row.numel() else 0
        shape = (n, n)
    return SparseMatrix(row, col, val, shape)


def from_csr(indptr, indices, val=None, shape=None):
    indptr = indptr.to(torch.int64)
    m = indptr.numel() - 1
    row = torch.repeat_interleave(torch.arange(m, device=indptr.device), indptr[1:] - indptr[:-1])
    col = indices.to(torch.int64)
    if val is None:
        val = torch.ones(col.numel(), dtype=torch.float32, device=col.device)
    return SparseMatrix(row, col, val, shape if shape is not None else (m, m))


def val_like(A, val):
    return SparseMatrix(A.row, A.col, val, A.shape)


def bsddmm(A, X1, X2):
    """Batched SDDMM: X1 [N, d, nh], X2 [d, M, nh] -> values [nnz, nh] = (X1 @ X2)[row, col] * A.val."""
    lhs = X1[A.row]                       # [nnz, d, nh]
    rhs = X2.permute(1, 0, 2)[A.col]      # [nnz, d, nh]
    v = (lhs * rhs).sum(1)
    a = A.val if A.val.dim() == 2 else A.val.unsqueeze(1)
    return SparseMatrix(A.row, A.col, v * a, A.shape)


def bspmm(A, X):
    """Batched SpMM: A values [nnz, nh], X [M, d, nh] -> [N, d, nh]."""
    v = A.val if A.val.dim() == 2 else A.val.unsqueeze(1)
    out = torch.zeros((A.shape[0],) + tuple(X.shape[1:]), dtype=X.dtype, device=X.device)
    return out.index_add_(0, A.row, X[A.col] * v.unsqueeze(1))
