"""Layer factory + graph preprocessing, without DGL.

Same function names, return tuples and `--format` dispatch as the reference's DFGNN/layers/util.py
(:52-57 g_to_SPmatrix, :66-100 preprocess_CSR/Hyper, :116-142 preprocess_Hyper_fw_bw,
:145-162 preprocess_softmax, :362-491 load_*).  COO -> CSR/CSC is done with torch sorts on whatever
device the graph lives on; index arrays come out int32, values fp32 (all ones, like
`dglsp.spmatrix`'s default).  Graphs that live on a CUDA device are converted by the native GPU preprocessing
(dfgnn_preprocess.coo_to_hyper -> include/dfgnn.h dfgnn_preprocess_hyper: two short radix sorts); the arrays are the
same either way (tests/test_gpu_parity.py::test_native_preprocess_matches_torch_path).

`smem_consume` keeps the reference's formula (128 neighbours/row guess) purely for signature
compatibility: the kernels size LDS per workgroup themselves and never overflow.
"""
import torch

from DFGNN.utils import sparse as dglsp

from .AGNN import (AGNNConv_csr, AGNNConv_csr_gm, AGNNConv_hyper, AGNNConv_softmax, AGNNConv_softmax_gm,
                   AGNNConv_tiling)
from .GAT_DOT import DOTGATConv_csr, DOTGATConv_hyper, DOTGATConv_softmax
from .GAT import (GATConv_dgNN, GATConv_hyper, GATConv_hyper_ablation, GATConv_hyper_recompute, GATConv_hyper_v2,
                  GATConv_softmax, GATConv_softmax_gm, GATConv_tiling)
from .GT import (SparseMHA_CSR, SparseMHA_CSR_GM, SparseMHA_forward_timing, SparseMHA_hyper, SparseMHA_softmax,
                 SparseMHA_softmax_gm, SparseMHA_tiling)

WARP_SIZE = 32  # only used by the smem_consume formula kept from the reference


def g_to_SPmatrix(g):
    indices = torch.stack(g.edges())
    N = g.num_nodes()
    return dglsp.spmatrix(indices, shape=(N, N)), 128


def _round_up(x, to):
    return (x + to - 1) // to * to


def _csr_parts(A):
    row_ptr, col_ind, val_idx = A.csr()
    return row_ptr.int(), col_ind.int(), A.val[val_idx]


def _unit_val(nnz, device):
    """A.val[val_idx] of an unweighted adjacency (g_to_SPmatrix: dglsp.spmatrix's default ones), without the
    gather; marked as all-ones for the binding (fused_gtconv val_ptr) so that it is not re-checked per batch."""
    val = torch.ones(nnz, dtype=torch.float32, device=device)
    val._dfgnn_unit = (val._version, True)
    return val


def _native(A, csc):
    import dfgnn_preprocess
    return dfgnn_preprocess.coo_to_hyper(A.row, A.col, A.shape[0], csc=csc)


def preprocess_CSR(g, **args):
    A, max_neigh = g_to_SPmatrix(g)
    if A.row.is_cuda:
        row_ptr, col_ind, _, _ = _native(A, csc=False)
        return row_ptr, col_ind, _unit_val(A.nnz, A.device), _round_up(max_neigh, WARP_SIZE)
    row_ptr, col_ind, val = _csr_parts(A)
    return row_ptr, col_ind, val, _round_up(max_neigh, WARP_SIZE)


def preprocess_Hyper(g, **args):
    A, max_neigh = g_to_SPmatrix(g)
    if A.row.is_cuda:
        row_ptr, col_ind, rows, _ = _native(A, csc=False)
        return row_ptr, col_ind, rows, _unit_val(A.nnz, A.device), _round_up(max_neigh * 8, WARP_SIZE)
    rows = torch.sort(A.row.int()).values
    row_ptr, col_ind, val = _csr_parts(A)
    return row_ptr, col_ind, rows, val, _round_up(max_neigh * 8, WARP_SIZE)


def preprocess_softmax(g, **args):
    A, max_neigh = g_to_SPmatrix(g)
    if A.row.is_cuda:
        row_ptr, col_ind, rows, _ = _native(A, csc=False)
        return row_ptr, col_ind, rows, _unit_val(A.nnz, A.device), _round_up(max_neigh, WARP_SIZE)
    rows = torch.sort(A.row.int()).values
    row_ptr, col_ind, val = _csr_parts(A)
    return row_ptr, col_ind, rows, val, _round_up(max_neigh, WARP_SIZE)


def preprocess_Hyper_fw_bw(g, fused=True):
    A, max_neigh = g_to_SPmatrix(g)
    if not fused:
        return A, None, None, None, None, None, None, None, None
    if A.row.is_cuda:
        row_ptr, col_ind, rows, _, col_ptr, row_ind, val_idx = _native(A, csc=True)
        return (A, rows, row_ptr, col_ind, _unit_val(A.nnz, A.device), col_ptr, row_ind, val_idx,
                _round_up(max_neigh * 8, WARP_SIZE))
    rows = torch.sort(A.row.int()).values
    row_ptr, col_ind, val = _csr_parts(A)
    A_csr = dglsp.from_csr(indptr=row_ptr, indices=col_ind, val=val, shape=A.shape)
    col_ptr, row_ind, val_idx = A_csr.csc()
    return (A, rows, row_ptr, col_ind, val, col_ptr.int(), row_ind.int(), val_idx.int(),
            _round_up(max_neigh * 8, WARP_SIZE))


_GT_LAYERS = {
    "csr": SparseMHA_CSR, "csr_gm": SparseMHA_CSR_GM, "tiling": SparseMHA_tiling, "hyper": SparseMHA_hyper,
    "nofuse": SparseMHA_hyper, "softmax": SparseMHA_softmax, "softmax_gm": SparseMHA_softmax_gm,
    "forward": SparseMHA_forward_timing,
    "hyper_ablation": SparseMHA_hyper,  # reference :385-386 (ablation entry; served by the production kernel)
}
_GAT_LAYERS = {
    "csr": GATConv_dgNN, "tiling": GATConv_tiling, "hyper": GATConv_hyper, "nofuse": GATConv_hyper,
    "softmax": GATConv_softmax, "softmax_gm": GATConv_softmax_gm,
    "hyper_v2": GATConv_hyper_v2, "hyper_recompute": GATConv_hyper_recompute, "hyper_ablation": GATConv_hyper_ablation,
}
_AGNN_LAYERS = {  # reference :424-442
    "hyper": AGNNConv_hyper, "csr": AGNNConv_csr, "softmax": AGNNConv_softmax, "csr_gm": AGNNConv_csr_gm,
    "tiling": AGNNConv_tiling, "softmax_gm": AGNNConv_softmax_gm,
}
_DOTGAT_LAYERS = {"hyper": DOTGATConv_hyper, "csr": DOTGATConv_csr, "softmax": DOTGATConv_softmax}
# formats of the reference that are baselines on NVIDIA-only libraries or paper experiments
_OUT_OF_SCOPE = {"hybrid", "pyg", "cugraph", "subgraph"}


def _pick(table, args, conv):
    if args.format in table:
        return table[args.format](args.dim, args.dim, args.heads)
    if args.format in _OUT_OF_SCOPE:
        raise ValueError(f"format {args.format} for {conv} is outside this build's scope (SURVEY.md 2.1)")
    raise ValueError(f"Unsupported format {args.format} in {conv}")


def load_layer_GT(args):
    return _pick(_GT_LAYERS, args, "GTconv")


def load_layer_GAT(args):
    return _pick(_GAT_LAYERS, args, "GATconv")


def load_layer_AGNN(args):
    return _pick(_AGNN_LAYERS, args, "AGNNconv")


def load_layer_DOTGAT(args):
    """The reference ships the DOTGAT layer classes (DFGNN/layers/GAT_DOT) without a `--conv` entry; `--conv dotgat` is
    this build's addition.  Its layers take (g, *preprocess tuple): load_prepfunc(args) wraps the graph in."""
    return _pick(_DOTGAT_LAYERS, args, "DOTGATconv")


def load_graphconv_layer(args):
    if args.conv == "dotgat":
        return load_layer_DOTGAT(args)
    if args.conv == "gat":
        return load_layer_GAT(args)
    if args.conv == "agnn":
        return load_layer_AGNN(args)
    if args.conv == "gt":
        return load_layer_GT(args)
    raise ValueError(f"unknown graph conv {args.conv}")


def load_prepfunc(args):
    if getattr(args, "conv", None) == "dotgat":
        inner = {"csr": preprocess_CSR, "hyper": preprocess_Hyper, "softmax": preprocess_softmax}[args.format]
        return lambda g, **kw: (g,) + tuple(inner(g, **kw))
    if args.format in ("csr", "csr_gm", "tiling"):
        return preprocess_CSR
    if args.format in ("hyper", "nofuse", "hyper_ablation", "hyper_recompute", "hyper_v2"):
        return preprocess_Hyper
    if args.format in ("softmax", "softmax_gm"):
        return preprocess_softmax
    if args.format == "forward":
        return preprocess_Hyper_fw_bw
    raise ValueError(f"Unsupported format {args.format}")
