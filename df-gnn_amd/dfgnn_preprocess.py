"""GPU-side graph preprocessing through the C ABI (include/dfgnn.h: dfgnn_preprocess_hyper).

COO edge list -> the CSR / sorted-COO-rows / CSC arrays of the reference's preprocess_Hyper and
preprocess_Hyper_fw_bw (DFGNN/layers/util.py:82-100, 116-142), for graphs that live on the GPU.  The reference gets
them from dgl.sparse (`A.csr()`, `torch.sort(A.row)`, `from_csr(...).csc()`); DFGNN/layers/util.py calls this module
instead when the graph is on a CUDA device (CPU graphs keep the torch restatement in DFGNN/utils/sparse.py).
"""
import torch

import dfgnn_native as _n
from _binding_util import check_device, stream_ptr


def coo_to_hyper(src, dst, num_nodes, csc=True):
    """(row_ptr, col_ind, rows, edge_order[, col_ptr, row_ind, val_idx]) as int32 device tensors.

    src / dst: int64 or int32 CUDA tensors of equal length (row = src, column = dst).  Stable: the CSR keeps the COO
    order inside a row, the CSC keeps the CSR order inside a column -- the same arrays as the torch path."""
    ext = _n.ext()
    if ext is not None and hasattr(ext, "preprocess_hyper") and src.is_cuda and dst.is_cuda:
        return tuple(ext.preprocess_hyper(src, dst, int(num_nodes), bool(csc)))
    check_device(src=src, dst=dst)
    if src.dtype != dst.dtype or src.dtype not in (torch.int64, torch.int32):
        raise RuntimeError(f"src / dst must both be int64 or int32, got {src.dtype} / {dst.dtype}")
    if src.dim() != 1 or src.shape != dst.shape:
        raise RuntimeError(f"src / dst must be 1-D and of equal length, got {tuple(src.shape)} / {tuple(dst.shape)}")
    src, dst = src.contiguous(), dst.contiguous()
    m, nnz, dev = int(num_nodes), src.numel(), src.device
    if nnz >= 2 ** 31 or m >= 2 ** 31:
        raise RuntimeError("graphs with 2^31 or more nodes / edges are not supported (int32 index arrays)")
    L = _n.lib()
    with torch.cuda.device(dev):
        i32 = dict(dtype=torch.int32, device=dev)
        row_ptr, col_ind = torch.empty(m + 1, **i32), torch.empty(nnz, **i32)
        rows, order = torch.empty(nnz, **i32), torch.empty(nnz, **i32)
        outs = [row_ptr, col_ind, rows, order]
        if csc:
            outs += [torch.empty(m + 1, **i32), torch.empty(nnz, **i32), torch.empty(nnz, **i32)]
        ws_bytes = int(L.dfgnn_preprocess_ws_bytes(m, nnz))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        ptrs = [t.data_ptr() for t in outs] + [None] * (7 - len(outs))
        _n.check(L.dfgnn_preprocess_hyper(m, nnz, src.data_ptr(), dst.data_ptr(), int(src.dtype == torch.int64), *ptrs,
                                          ws.data_ptr(), ws_bytes, stream_ptr(dev)), "dfgnn_preprocess_hyper")
    return tuple(outs)
