"""`fused_gatconv` -- drop-in for the reference's extension module of the same name
(PYBIND11_MODULE(fused_gatconv), DFGNN/src/fused_gatconv/fused_gatconv.cpp:355-372), bound to the
MI355X HIP library through its C ABI (include/dfgnn.h).  See fused_gtconv.py for the conventions.

In scope: the four inference entry points of the hyper / softmax /
softmax_gm / tiling variants, and the training pair gat_forward / gat_backward
behind FusedGATFunction and (8f rank 3) the hyper_v2 / hyper_recompute entry points of the reference's comparison
sweeps.  gat_forward_tb (the reference's tile-scheduler experiment, no Python caller there) returns the same three
tensors from the training forward; its schedule argument only distributes work and is validated, not followed.
"""
import torch

import dfgnn_native as _n
from _binding_util import (_KeyedCache, as_int32, check_contiguous, check_device, check_dtype, get_plan, get_rows, ptr,
                           stream_ptr)

# Set to False to force the general (plan-less) kernels; results are identical either way.
USE_BLOCK_PLAN = True


def _check(attn_row, attn_col, indptr, indices, rows, in_feat):
    tensors = dict(attn_row=attn_row, attn_col=attn_col, indptr=indptr, indices=indices, in_feat=in_feat)
    if rows is not None:
        tensors["rows"] = rows
    check_device(**tensors)
    check_contiguous(**tensors)
    check_dtype(torch.float32, attn_row=attn_row, attn_col=attn_col, in_feat=in_feat)
    check_dtype(torch.int32, indptr=indptr, indices=indices)
    if rows is not None:
        check_dtype(torch.int32, rows=rows)
    if in_feat.dim() != 3:
        raise RuntimeError(f"in_feat must have shape [nodes, heads, feat], got {tuple(in_feat.shape)}")
    m, nnz = indptr.size(0) - 1, indices.size(0)
    h, f = attn_row.size(1) if attn_row.dim() == 2 else -1, in_feat.size(2)
    if tuple(attn_row.shape) != (m, in_feat.size(1)) or tuple(attn_col.shape) != (m, in_feat.size(1)):
        raise RuntimeError(f"attn_row / attn_col must have shape ({m}, {in_feat.size(1)}), got "
                           f"{tuple(attn_row.shape)} / {tuple(attn_col.shape)}")
    if in_feat.size(0) != m:
        raise RuntimeError(f"indptr describes {m} rows but in_feat has {in_feat.size(0)} nodes")
    if rows is not None and (rows.dim() != 1 or rows.size(0) != nnz):
        raise RuntimeError(f"rows must have shape ({nnz},), got {tuple(rows.shape)}")
    return m, nnz, h, f


def gat_inference_hyper(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """fused_gatconv.cpp:99-119 -> Tensor out[m, h, f]"""
    ext = _n.ext()
    if ext is not None:  # torch C++ binding (csrc/torch_ext.cpp): same checks, same C ABI call
        plan, meta, need_ws = get_plan(indptr, indices, in_feat.size(-1) if in_feat.dim() == 3 else 0, USE_BLOCK_PLAN)
        return ext.gat_hyper_fwd(attn_row, attn_col, indptr, indices, rows, float(negative_slope), in_feat, plan or 0,
                                 meta or 0, need_ws)
    m, nnz, h, f = _check(attn_row, attn_col, indptr, indices, rows, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty_like(in_feat)
        plan, meta, need_ws = get_plan(indptr, indices, f, USE_BLOCK_PLAN)
        ws = torch.empty((h, nnz), dtype=torch.float32, device=in_feat.device) if need_ws else None
        _n.check(_n.lib().dfgnn_gat_hyper_fwd(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(rows), ptr(attn_row),
                                              ptr(attn_col), float(negative_slope), ptr(in_feat), ptr(ws), ptr(out),
                                              plan, meta, stream_ptr(in_feat.device)), "gat_inference_hyper")
    return out


def gat_inference_hyper_ablation(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope,
                                 in_feat):
    """fused_gatconv.cpp (ablation study entry): served by the production kernel."""
    return gat_inference_hyper(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat)


def _softmax(fn_name, what, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    ext = _n.ext()
    if ext is not None:
        return ext.gat_softmax_fwd(attn_row, attn_col, indptr, indices, rows, float(negative_slope), in_feat,
                                   fn_name == "dfgnn_gat_softmax_fwd")
    m, nnz, h, f = _check(attn_row, attn_col, indptr, indices, rows, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty_like(in_feat)
        logits = torch.empty((h, nnz), dtype=torch.float32, device=in_feat.device)
        _n.check(getattr(_n.lib(), fn_name)(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(rows), ptr(attn_row),
                                            ptr(attn_col), float(negative_slope), ptr(in_feat), ptr(logits),
                                            ptr(out), stream_ptr(in_feat.device)), what)
    return out


def gat_inference_softmax(smem_consume, attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """fused_gatconv.cpp:40-61 -> Tensor"""
    return _softmax("dfgnn_gat_softmax_fwd", "gat_inference_softmax", attn_row, attn_col, indptr, indices, rows,
                    negative_slope, in_feat)


def gat_inference_softmax_gm(attn_row, attn_col, indptr, indices, rows, negative_slope, in_feat):
    """fused_gatconv.cpp:69-90 -> Tensor"""
    return _softmax("dfgnn_gat_softmax_gm_fwd", "gat_inference_softmax_gm", attn_row, attn_col, indptr, indices,
                    rows, negative_slope, in_feat)


# ---- 'tiling' on super-node full graphs: column chunks that fit an XCD's L2 (csrc/gat_tiling_chunked.hip) -----------------
# The chunk-major edge order of a graph is preprocessing, like the block plan of a batch: built on first use, cached by
# the identity of the CSR arrays' memory.  TILING_CHUNK_ROWS = 0 keeps the single-kernel form everywhere.
# Rows of a chunk: TILING_CHUNK_BYTES / (4 f), a multiple of 1024, at most 32768 (16-bit column offsets).  6 MiB per chunk
# measured best on the reddit-like graph at f = 128 (12288 rows: 4.28 ms; 8192 rows = one XCD's 4 MiB L2: 4.89 ms -- more
# chunks, more partial states; 16384 rows: 4.74 ms; the single-kernel form: 8.27 ms).  0 = the single-kernel form everywhere.
TILING_CHUNK_BYTES = 6 << 20
TILING_CHUNK_ROWS = None           # tests: a fixed number of rows per chunk instead
TILING_CHUNK_MIN_TABLE = 64 << 20  # feature table (bytes) below which the L2s hold it anyway
TILING_CHUNK_MIN_DEGREE = 64       # average degree below which a row of X is not re-used enough to pay for the partial states
_chunk_cache = _KeyedCache(entries=4)


def _tiling_chunks(row_ptr, col_ind, chunk_rows):
    """(seg_ptr int32[nchunks m + 1], ccol int16[nnz]) of include/dfgnn.h: dfgnn_gat_tiling_chunked_fwd."""
    key = _KeyedCache.key_of(row_ptr, col_ind, extra=(chunk_rows,))
    hit = _chunk_cache.get(key)
    if hit is None:
        m, nnz = row_ptr.size(0) - 1, col_ind.size(0)
        nchunks = (m + chunk_rows - 1) // chunk_rows
        with torch.cuda.device(col_ind.device):
            chunk = torch.div(col_ind, chunk_rows, rounding_mode="floor").to(torch.int16)
            order = torch.sort(chunk, stable=True).indices           # CSR (row-major) order survives inside a chunk
            ccol = (col_ind - chunk.to(torch.int32) * chunk_rows).to(torch.int16)[order].contiguous()
            rows = get_rows(row_ptr, nnz)
            counts = torch.bincount(chunk.long() * m + rows.long(), minlength=nchunks * m)
            seg_ptr = torch.zeros(nchunks * m + 1, dtype=torch.int32, device=col_ind.device)
            seg_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
            del chunk, order, counts
        hit = _chunk_cache.put(key, (seg_ptr, ccol), row_ptr, col_ind)
    return hit


def _chunk_rows(f):
    if TILING_CHUNK_ROWS is not None:
        return int(TILING_CHUNK_ROWS)
    return min(32768, (TILING_CHUNK_BYTES // (4 * f)) // 1024 * 1024)


def _use_chunked_tiling(m, nnz, h, f):
    return (_chunk_rows(f) > 0 and m * h * f * 4 >= TILING_CHUNK_MIN_TABLE and nnz >= TILING_CHUNK_MIN_DEGREE * m and
            nnz < 2 ** 31)


def gat_inference_tiling(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """fused_gatconv.cpp:196-219 -> Tensor"""
    if in_feat.dim() == 3 and in_feat.is_cuda and row_ptr.dim() == 1 and col_ind.dim() == 1 and \
            _use_chunked_tiling(row_ptr.size(0) - 1, col_ind.size(0), in_feat.size(1), in_feat.size(2)):
        m, nnz, h, f = _check(attn_row, attn_col, row_ptr, col_ind, None, in_feat)
        chunk_rows = _chunk_rows(f)
        seg_ptr, ccol = _tiling_chunks(row_ptr, col_ind, chunk_rows)
        L = _n.lib()
        with torch.cuda.device(in_feat.device):
            out = torch.empty_like(in_feat)
            ws_bytes = int(L.dfgnn_gat_tiling_chunked_ws_bytes(m, h, f, chunk_rows))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=in_feat.device)
            _n.check(L.dfgnn_gat_tiling_chunked_fwd(m, nnz, h, f, chunk_rows, ptr(seg_ptr), ptr(ccol), ptr(attn_row),
                                                    ptr(attn_col), float(negative_slope), ptr(in_feat), ptr(out), ptr(ws),
                                                    ws_bytes, stream_ptr(in_feat.device)), "gat_inference_tiling (chunked)")
        return out
    ext = _n.ext()
    if ext is not None:
        return ext.gat_tiling_fwd(attn_row, attn_col, row_ptr, col_ind, float(negative_slope), in_feat)
    m, nnz, h, f = _check(attn_row, attn_col, row_ptr, col_ind, None, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty_like(in_feat)
        _n.check(_n.lib().dfgnn_gat_tiling_fwd(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(attn_row),
                                               ptr(attn_col), float(negative_slope), ptr(in_feat), ptr(out),
                                               stream_ptr(in_feat.device)), "gat_inference_tiling")
    return out


def gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """fused_gatconv.cpp (dgNN node-parallel CSR inference): same function as the
    tiling kernel, which serves it."""
    return gat_inference_tiling(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat)


def gat_inference_hyper_recompute(attn_row, attn_col, indptr, indices, negative_slope, in_feat):
    """fused_gatconv.cpp:124-142 -> Tensor.  CSR only; node-parallel, no logit storage: the rank-one logits are recomputed
    in each of the three sweeps (max, sum, weighted sum), like the reference's kernel
    (fused_gatconv_hyper_recompute.cu:118-216) -- csrc/csr_fwd.hip:gat_recompute_fwd_kernel; any f (the reference exit(0)s
    unless f % 128 == 0)."""
    m, nnz, h, f = _check(attn_row, attn_col, indptr, indices, None, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty_like(in_feat)
        _n.check(_n.lib().dfgnn_gat_recompute_fwd(m, nnz, h, f, ptr(indptr), ptr(indices), ptr(attn_row), ptr(attn_col),
                                                  float(negative_slope), ptr(in_feat), ptr(out),
                                                  stream_ptr(in_feat.device)), "gat_inference_hyper_recompute")
    return out


def gat_inference_hyper_v2(smem_consume, a_l, a_r, indptr, indices, negative_slope, in_feat):
    """fused_gatconv.cpp:148-158 -> Tensor.  a_l, a_r: the layer's attention vectors, [heads, f] (a leading 1 is
    accepted, as the layers pass them).  Two kernels like the reference's (fused_gatconv_hyper_v2.cu:251-281): the
    per-node scores <a_l, X_i>, <a_r, X_i> in one pass over X, then the fused 'hyper' convolution -- with a block plan
    and the COO rows derived from indptr (this entry point takes CSR only), else the CSR tiling kernel.
    The reference reads a_l / a_r as contiguous [heads, f] whatever their strides (the layers hand it a transposed
    view: only right for heads == 1); here they are made contiguous first."""
    check_device(a_l=a_l, a_r=a_r, in_feat=in_feat)
    check_dtype(torch.float32, a_l=a_l, a_r=a_r, in_feat=in_feat)
    if in_feat.dim() != 3:
        raise RuntimeError(f"in_feat must have shape [nodes, heads, feat], got {tuple(in_feat.shape)}")
    m, h, f = in_feat.shape
    a_l, a_r = a_l.reshape(-1, a_l.shape[-1]).contiguous(), a_r.reshape(-1, a_r.shape[-1]).contiguous()
    if tuple(a_l.shape) != (h, f) or tuple(a_r.shape) != (h, f):
        raise RuntimeError(f"a_l / a_r must have shape ({h}, {f}), got {tuple(a_l.shape)} / {tuple(a_r.shape)}")
    check_contiguous(in_feat=in_feat)
    dev = in_feat.device
    with torch.cuda.device(dev):
        attn_row = torch.empty((m, h), dtype=torch.float32, device=dev)
        attn_col = torch.empty((m, h), dtype=torch.float32, device=dev)
        _n.check(_n.lib().dfgnn_gat_attn_scores(m, h, f, ptr(a_l), ptr(a_r), ptr(in_feat), ptr(attn_row), ptr(attn_col),
                                                stream_ptr(dev)), "gat_inference_hyper_v2 (scores)")
    _check(attn_row, attn_col, indptr, indices, None, in_feat)
    plan, _, _ = get_plan(indptr, indices, f, USE_BLOCK_PLAN)
    if plan is not None:
        return gat_inference_hyper(smem_consume, attn_row, attn_col, indptr, indices, get_rows(indptr, indices.size(0)),
                                   negative_slope, in_feat)
    return gat_inference_tiling(attn_row, attn_col, indptr, indices, negative_slope, in_feat)


def _train_plan(row_ptr, col_ind, f, attn_drop):
    """(rows, plan, meta) for the training pair: the block plan and the COO row ids when the batch may qualify for
    the matrix-core kernels (the library checks that every range of the plan is dense), else Nones."""
    plan, meta, _ = get_plan(row_ptr, col_ind, f, USE_BLOCK_PLAN)
    if plan is None:
        return None, None, None
    return get_rows(row_ptr, col_ind.size(0)), plan, meta


def gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop):
    """fused_gatconv.cpp:11-32 -> [out_feat[m,h,f], edge_max[m,h], edge_sum[m,h], edge_mask[nnz,h]]

    edge_mask holds the uniform randoms of the attention dropout (the reference fills it with cuRAND seeded by
    clock(), fused_gatconv_kernel.cu:1074-1083; here torch.rand, so torch.manual_seed reproduces a run).  With
    attn_drop == 0 nothing is dropped and no randoms are drawn: edge_mask is then a stride-0 view of a single 1.0
    (same shape, no memory), which gat_backward accepts for attn_drop == 0."""
    ext = _n.ext()
    if ext is not None and hasattr(ext, "gat_fwd_train") and in_feat.dim() == 3 and in_feat.is_cuda:
        # torch C++ binding (csrc/torch_ext.cpp): same checks, same C ABI call, ~5 us of host time instead of ~50
        attn_drop = float(attn_drop)
        nnz_, h_ = col_ind.size(0), in_feat.size(1)
        with torch.cuda.device(in_feat.device):
            if attn_drop > 0.0:
                edge_mask = torch.rand((nnz_, h_), dtype=torch.float32, device=in_feat.device)
            else:
                edge_mask = torch.ones((1, 1), dtype=torch.float32, device=in_feat.device).expand(nnz_, h_)
        rows, plan, meta = _train_plan(row_ptr, col_ind, in_feat.size(2), attn_drop)
        out, edge_max, edge_sum = ext.gat_fwd_train(attn_row, attn_col, row_ptr, col_ind, rows, float(negative_slope), in_feat,
                                                    edge_mask if attn_drop > 0.0 else None, attn_drop, plan or 0, meta or 0)
        return [out, edge_max, edge_sum, edge_mask]
    m, nnz, h, f = _check(attn_row, attn_col, row_ptr, col_ind, None, in_feat)
    attn_drop = float(attn_drop)
    if not 0.0 <= attn_drop < 1.0:
        raise RuntimeError(f"attn_drop must be in [0, 1), got {attn_drop}")
    dev = in_feat.device
    with torch.cuda.device(dev):
        out = torch.empty_like(in_feat)
        edge_max = torch.empty((m, h), dtype=torch.float32, device=dev)
        edge_sum = torch.empty((m, h), dtype=torch.float32, device=dev)
        if attn_drop > 0.0:
            edge_mask = torch.rand((nnz, h), dtype=torch.float32, device=dev)
            mask_ptr = ptr(edge_mask)
        else:
            edge_mask = torch.ones((1, 1), dtype=torch.float32, device=dev).expand(nnz, h)
            mask_ptr = None
        rows, plan, meta = _train_plan(row_ptr, col_ind, f, attn_drop)
        _n.check(_n.lib().dfgnn_gat_fwd_train(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(rows), ptr(attn_row),
                                              ptr(attn_col), float(negative_slope), ptr(in_feat), mask_ptr,
                                              attn_drop, ptr(edge_max), ptr(edge_sum), ptr(out), plan, meta,
                                              stream_ptr(dev)), "gat_forward")
    return [out, edge_max, edge_sum, edge_mask]


def gat_backward(negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                 edge_mask, in_feat, attn_row, attn_col, grad):
    """fused_gatconv.cpp:291-353 -> [grad_feat[m,h,f], grad_attn_row[m,h], grad_attn_col[m,h]]"""
    ext = _n.ext()
    if ext is not None and hasattr(ext, "gat_bwd") and in_feat.dim() == 3 and in_feat.is_cuda:
        attn_drop = float(attn_drop)
        rows, plan, meta = _train_plan(row_ptr, col_ind, in_feat.size(2), attn_drop)
        return ext.gat_bwd(float(negative_slope), attn_drop, row_ptr, col_ind, rows, col_ptr, row_ind, as_int32(permute),
                           edge_max, edge_sum, edge_mask if attn_drop > 0.0 else None, in_feat, attn_row, attn_col, grad,
                           plan or 0, meta or 0)
    m, nnz, h, f = _check(attn_row, attn_col, row_ptr, col_ind, None, in_feat)
    attn_drop = float(attn_drop)
    if not 0.0 <= attn_drop < 1.0:
        raise RuntimeError(f"attn_drop must be in [0, 1), got {attn_drop}")
    permute = as_int32(permute)      # (dgl hands the CSC -> CSR permutation over as int64, like GT's val_idx)
    tensors = dict(col_ptr=col_ptr, row_ind=row_ind, permute=permute, edge_max=edge_max, edge_sum=edge_sum,
                   grad=grad)
    check_device(edge_mask=edge_mask, **tensors)
    check_contiguous(**tensors)
    check_dtype(torch.int32, col_ptr=col_ptr, row_ind=row_ind, permute=permute)
    check_dtype(torch.float32, edge_max=edge_max, edge_sum=edge_sum, edge_mask=edge_mask, grad=grad)
    if grad.shape != in_feat.shape:
        raise RuntimeError(f"grad has shape {tuple(grad.shape)}, expected {tuple(in_feat.shape)}")
    if tuple(edge_max.shape) != (m, h) or tuple(edge_sum.shape) != (m, h):
        raise RuntimeError(f"edge_max / edge_sum must have shape ({m}, {h})")
    if col_ptr.size(0) != m + 1 or row_ind.size(0) != nnz or permute.size(0) != nnz:
        raise RuntimeError("col_ptr / row_ind / permute do not match the CSR structure")
    mask_ptr = None
    if attn_drop > 0.0:
        if tuple(edge_mask.shape) != (nnz, h):
            raise RuntimeError(f"edge_mask must have shape ({nnz}, {h}), got {tuple(edge_mask.shape)}")
        check_contiguous(edge_mask=edge_mask)
        mask_ptr = ptr(edge_mask)
    dev = in_feat.device
    with torch.cuda.device(dev):
        grad_feat = torch.empty_like(in_feat)
        grad_attn_row = torch.empty((m, h), dtype=torch.float32, device=dev)
        grad_attn_col = torch.empty((m, h), dtype=torch.float32, device=dev)
        grad_edge = torch.empty((h, nnz), dtype=torch.float32, device=dev)
        rows, plan, meta = _train_plan(row_ptr, col_ind, f, attn_drop)
        _n.check(_n.lib().dfgnn_gat_bwd(m, nnz, h, f, ptr(row_ptr), ptr(col_ind), ptr(rows), ptr(col_ptr),
                                        ptr(row_ind), ptr(permute), ptr(attn_row), ptr(attn_col),
                                        float(negative_slope), ptr(in_feat), ptr(edge_max), ptr(edge_sum), mask_ptr,
                                        attn_drop, ptr(grad), ptr(grad_edge), ptr(grad_feat), ptr(grad_attn_row),
                                        ptr(grad_attn_col), plan, meta, stream_ptr(dev)), "gat_backward")
    return [grad_feat, grad_attn_row, grad_attn_col]


def gat_forward_tb(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, tile_scheduler):
    """fused_gatconv.cpp:256-282 -> [out_feat[m,h,f], edge_max[m,h], edge_sum[m,h]].

    The reference's kernel (fused_gatconv_kernel.cu:976-1060) walks `tile_scheduler` -- int32 (row, 32-edge tile of the
    row) pairs -- one workgroup per entry, and combines the tiles of a row through atomics on edge_max / edge_sum with
    no grid-wide synchronisation (its result depends on workgroup timing).  The schedule only says who computes what;
    the values it is meant to produce are those of gat_forward without dropout, which is what this returns (the MI355X
    kernels split heavy rows themselves)."""
    check_device(tile_scheduler=tile_scheduler, in_feat=in_feat)
    check_contiguous(tile_scheduler=tile_scheduler)
    check_dtype(torch.int32, tile_scheduler=tile_scheduler)
    out, edge_max, edge_sum, _ = gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, 0.0)
    return [out, edge_max, edge_sum]
