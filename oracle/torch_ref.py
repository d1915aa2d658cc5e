"""Second, independently written restatement of the reference's non-fused path, in torch.

TEST INFRASTRUCTURE ONLY (see oracle/oracle.c).  PARITY UNPINNED (same reason).

Where ``oracle.c`` walks CSR rows with explicit loops and hand-derived gradient formulas, this
file restates ``forward_dglsp`` the way the reference layers call dgl.sparse -- as edge-list
tensor algebra -- and lets ``torch.autograd`` produce the gradients:

  GT   DFGNN/layers/GT/gtconv_layer.py:29-33    attn = bsddmm(A, q, k^T); attn.softmax(); bspmm(attn, v)
  GAT  DFGNN/layers/GAT/gatconv_layer.py:30-38  e = e_l[row] + e_r[col]; LeakyReLU; softmax; bspmm

The two agreeing (tests/test_oracle.py) is what stands in for golden vectors.
All functions work at the operator boundary: features ``[m, h, f]``, COO ``row``/``col`` int64.
"""
import torch


def _segment_softmax(logit, row, m):
    """Row-wise softmax over edge logits ``[nnz, h]`` grouped by ``row`` (dgl.sparse .softmax())."""
    h = logit.shape[1]
    idx = row.unsqueeze(1).expand(-1, h)
    mx = torch.full((m, h), float("-inf"), dtype=logit.dtype).scatter_reduce(
        0, idx, logit, reduce="amax", include_self=True)
    ex = torch.exp(logit - mx[row])
    den = torch.zeros((m, h), dtype=logit.dtype).index_add_(0, row, ex)
    return ex / den[row]


def gt_conv(row, col, val, Q, K, V):
    """out[m,h,f], attn[nnz,h] for the GT conv; differentiable in Q, K, V."""
    m = Q.shape[0]
    logit = (Q[row] * K[col]).sum(-1) * val.unsqueeze(1)      # bsddmm(A, q, k^T): (QK^T) * A
    attn = _segment_softmax(logit, row, m)                    # attn.softmax()
    out = torch.zeros_like(Q).index_add_(0, row, attn.unsqueeze(-1) * V[col])  # bspmm(attn, v)
    return out, attn


def gat_conv(row, col, attn_row, attn_col, negative_slope, X):
    """out[m,h,f], attn[nnz,h] for the GAT conv (attn_row/attn_col are [m,h])."""
    m = X.shape[0]
    negative_slope = float(torch.tensor(negative_slope, dtype=torch.float32))  # the C ABI takes a float
    e = attn_row[row] + attn_col[col]
    a = torch.nn.functional.leaky_relu(e, negative_slope)
    attn = _segment_softmax(a, row, m)
    out = torch.zeros_like(X).index_add_(0, row, attn.unsqueeze(-1) * X[col])
    return out, attn


def csr_to_coo(indptr, indices):
    indptr = torch.as_tensor(indptr, dtype=torch.int64)
    deg = indptr[1:] - indptr[:-1]
    row = torch.repeat_interleave(torch.arange(len(deg), dtype=torch.int64), deg)
    return row, torch.as_tensor(indices, dtype=torch.int64)


def gt_forward_backward(indptr, indices, val, Q, K, V, dO=None, dtype=torch.float64):
    """Convenience: CSR in, (out, attn[h,nnz]) and optionally (dQ, dK, dV) out, all in ``dtype``."""
    row, col = csr_to_coo(indptr, indices)
    val = torch.as_tensor(val, dtype=dtype)
    Q, K, V = (torch.as_tensor(x).to(dtype).clone().requires_grad_(dO is not None) for x in (Q, K, V))
    out, attn = gt_conv(row, col, val, Q, K, V)
    if dO is None:
        return out.detach(), attn.detach().t().contiguous()
    out.backward(torch.as_tensor(dO).to(dtype))
    return out.detach(), attn.detach().t().contiguous(), Q.grad, K.grad, V.grad


def gat_forward(indptr, indices, attn_row, attn_col, negative_slope, X, dtype=torch.float64):
    row, col = csr_to_coo(indptr, indices)
    out, attn = gat_conv(row, col, torch.as_tensor(attn_row).to(dtype), torch.as_tensor(attn_col).to(dtype),
                         negative_slope, torch.as_tensor(X).to(dtype))
    return out, attn.t().contiguous()


def gat_train(indptr, indices, attn_row, attn_col, negative_slope, X, dO=None, edge_mask=None, attn_drop=0.0,
              dtype=torch.float64):
    """GAT training pair as edge-list algebra + autograd: softmax, then (optionally) attention dropout with the
    given uniform randoms ``edge_mask[nnz, h]`` (keep where mask > attn_drop, scale 1/(1-attn_drop);
    DFGNN/src/fused_gatconv/fused_gatconv_kernel.cu:101-110), then spmm.  Returns ``out`` and, when ``dO`` is
    given, ``(grad_feat, grad_attn_row, grad_attn_col)``."""
    row, col = csr_to_coo(indptr, indices)
    ar, ac, Xt = (torch.as_tensor(x).to(dtype).clone().requires_grad_(dO is not None)
                  for x in (attn_row, attn_col, X))
    slope = float(torch.tensor(negative_slope, dtype=torch.float32))
    a = torch.nn.functional.leaky_relu(ar[row] + ac[col], slope)
    attn = _segment_softmax(a, row, Xt.shape[0])
    if edge_mask is not None:
        drop = float(torch.tensor(attn_drop, dtype=torch.float32))
        keep = (torch.as_tensor(edge_mask, dtype=torch.float32) > drop).to(dtype)
        attn = attn * keep / (1.0 - drop)
    out = torch.zeros_like(Xt).index_add_(0, row, attn.unsqueeze(-1) * Xt[col])
    if dO is None:
        return out.detach()
    out.backward(torch.as_tensor(dO).to(dtype))
    return out.detach(), Xt.grad, ar.grad, ac.grad
