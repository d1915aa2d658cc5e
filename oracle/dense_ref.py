"""oracle/dense_ref.py -- third restatement of the path, as masked DENSE attention in float64 numpy.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c): a cross-check of the C oracle and of oracle/torch_ref.py, written against
the math spec rather than against either of them: per head, S = Q K^T (GT) or the rank-one
LeakyReLU(attn_row + attn_col^T) (GAT) as a full m x m matrix, an edge-count matrix C (C[i, j] = how often (i, j) occurs:
duplicate edges are separate softmax terms, DFGNN/layers/util.py:53-56), P = C exp(S - rowmax) / rowsum, O = P V, and
the backward by the closed-form softmax Jacobian.  Quadratic in the node count: small graphs only.  Edge values must be 1
(a dense mask cannot carry per-edge weights of duplicate edges)."""
import numpy as np


def _counts(indptr, indices, m):
    C = np.zeros((m, m))
    for i in range(m):
        for j in indices[indptr[i]:indptr[i + 1]]:
            C[i, j] += 1.0
    return C


def _masked_softmax(S, C):
    has = C.sum(1) > 0
    Sm = np.where(C > 0, S, -np.inf)
    mx = np.where(has, Sm.max(1, initial=-np.inf), 0.0)
    E = C * np.exp(np.where(C > 0, S - mx[:, None], -np.inf))
    den = E.sum(1)
    return np.where(has[:, None], E / np.where(den > 0, den, 1.0)[:, None], 0.0)   # empty row -> 0 (hyper.cu:143)


def gt_forward_backward(indptr, indices, Q, K, V, dO):
    """-> out, dQ, dK, dV (float64 [m, h, f]); P per pair already sums a duplicate edge's equal terms."""
    m, h, f = Q.shape
    C = _counts(indptr, indices, m)
    out, dQ, dK, dV = (np.zeros((m, h, f)) for _ in range(4))
    for hd in range(h):
        q, k, v, g = (a[:, hd].astype(np.float64) for a in (Q, K, V, dO))
        P = _masked_softmax(q @ k.T, C)
        out[:, hd] = P @ v
        dP = g @ v.T
        dS = P * (dP - (P * dP).sum(1, keepdims=True))
        dQ[:, hd], dK[:, hd], dV[:, hd] = dS @ k, dS.T @ q, P.T @ g
    return out, dQ, dK, dV


def gat_forward(indptr, indices, attn_row, attn_col, slope, X):
    m, h, f = X.shape
    C = _counts(indptr, indices, m)
    out = np.zeros((m, h, f))
    for hd in range(h):
        S = attn_row[:, hd].astype(np.float64)[:, None] + attn_col[:, hd].astype(np.float64)[None, :]
        S = np.where(S > 0, S, slope * S)
        out[:, hd] = _masked_softmax(S, C) @ X[:, hd].astype(np.float64)
    return out
