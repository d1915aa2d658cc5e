"""GPU parity tests of the statistics-saving GT training pair (include/dfgnn.h: dfgnn_gt_hyper_fwd_stats /
dfgnn_gt_bwd_stats; csrc/gt_dense_stats.hip): forward without attn_edge, backward that recomputes the attention on the
matrix cores from two floats per (row, head).  Everything is compared with the CPU oracle (oracle/oracle.c) on identical
inputs, through the Python binding -> C ABI; tolerance as in test_gpu_parity.py (1e-3, in practice ~1e-6)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL = RTOL = 1e-3
DEV = "cuda:0"


def _close(got, want, what):
    got = got.detach().cpu().double().numpy()
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want)
    bad = err > ATOL + RTOL * np.abs(want)
    assert not bad.any(), f"{what}: {bad.sum()} elements off, max abs err {err.max():.3e}"
    return float(err.max()) if err.size else 0.0


def _geometry_batch(seed, duplicate=False):
    """Every code path of the matrix-core kernels: tiny graphs, <= 128 nodes, 129-160, 161-255, isolated nodes (empty
    rows AND columns), and two DIRECTED graphs (the out-edge bitmap differs from the in-edge bitmap; rows without
    out-edges, columns without in-edges)."""
    from DFGNN.utils import Graph, batch
    rng = np.random.default_rng(seed)

    def er(n, p, drop=()):
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(len(iu)) < p
        keep &= ~np.isin(iu, drop) & ~np.isin(ju, drop)
        return np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]])

    graphs = [Graph(*er(n, p), n) for n, p in ((9, 0.9), (17, 0.6), (64, 0.5), (128, 0.35), (129, 0.3), (145, 0.4), (160, 0.3),
                                               (161, 0.3), (200, 0.25), (255, 0.2))]
    graphs.append(Graph(*er(70, 0.5, drop=(0, 33, 69)), 70))
    for shape, p in (((80, 140), 0.3), ((150, 150), 0.2)):
        ds_, dd_ = np.nonzero(rng.random(shape) < p)
        graphs.append(Graph(ds_.astype(np.int64), dd_.astype(np.int64), max(shape)))
    if duplicate:
        s_, d_ = er(40, 0.6)
        graphs.append(Graph(np.concatenate([s_, s_[:1]]), np.concatenate([d_, d_[:1]]), 40))
    return batch(graphs).to(DEV)


def _row_stats(row_ptr, col_ind, Q, K):
    """float64 logit maximum and sum of exponentials per (row, head): what the forward must save."""
    m, h, _ = Q.shape
    mx = np.full((m, h), -1e38)
    sm = np.zeros((m, h))
    for i in range(m):
        cols = col_ind[row_ptr[i]:row_ptr[i + 1]]
        if len(cols):
            s = np.einsum("hf,jhf->jh", Q[i].astype(np.float64), K[cols].astype(np.float64))
            mx[i] = s.max(axis=0)
            sm[i] = np.exp(s - mx[i]).sum(axis=0)
    return mx, sm


@pytest.mark.parametrize("h,f", [(1, 128), (1, 64), (2, 64), (4, 32), (8, 16), (3, 32), (2, 8), (2, 128), (16, 16)])
def test_stats_pair_every_geometry(oracle_mod, h, f):
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = _geometry_batch(17 + f)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, h, f, seed=5, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    assert gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q)
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    dQ, dK, dV = gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V))
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    _close(out, want, "out")
    _close(dQ, wq, "dQ")
    _close(dK, wk, "dK")
    _close(dV, wv, "dV")
    wmx, wsm = _row_stats(n_(row_ptr), n_(col_ind), n_(Q), n_(K))
    _close(mx, wmx.astype(np.float32), "row_max")      # (-1e38 for empty rows on both sides)
    _close(sm, wsm, "row_sum")
    # the attn_edge pair (the reference's form) on the same inputs: same results to fp32 rounding
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out_a, attn = gt.gt_hyper_forward(*args)
    dQ_a, dK_a, dV_a = gt.gt_backward(*args, attn, dO)
    for a, b, what in ((out, out_a, "out"), (dQ, dQ_a, "dQ"), (dK, dK_a, "dK"), (dV, dV_a, "dV")):
        assert torch.allclose(a, b, atol=2e-5, rtol=1e-4), what


@pytest.mark.parametrize("h,f", [(1, 128), (1, 64), (2, 64), (4, 32), (8, 16), (3, 32), (2, 8), (2, 128)])
def test_weighted_edges_on_the_matrix_cores(oracle_mod, h, f):
    """Edge values other than ones (the reference's `attn * val`, fused_gtconv_hyper.cu:88-90): the statistics pair reads
    them in the plan's dense form (csrc/gt_dense_stats_w.hip) -- forward, inference and backward against the oracle on
    every range class, positive and negative values, and the dense form itself against the CSR arrays."""
    import fused_gtconv as gt
    from _binding_util import get_plan_obj, plan_dense_weights, val_ptr
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = _geometry_batch(29 + f)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    val = (torch.rand(nnz, generator=torch.Generator().manual_seed(f)) * 2.5 - 1.0).to(DEV)   # in [-1, 1.5)
    assert val_ptr(val) is not None
    Q, K, V = S.gt_features(m, h, f, seed=5, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    plan = gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q)
    assert plan is not None and plan.num_dense == plan.num_fit > 0 and gt.gt_stats_pair_chosen(row_ptr, col_ind, val, Q) is plan
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    # the dense form: W[256 i + (j - n0)] = val[e], zero elsewhere
    W = n_(plan_dense_weights(plan, row_ptr, val)).reshape(m, 256)
    rp, ci, vv = n_(row_ptr), n_(col_ind), n_(val).reshape(-1)
    fit = n_(plan.buf)[12:12 + 2 * plan.num_fit].reshape(-1, 2)
    want_w = np.zeros((m, 256), np.float32)
    for n0, n1f in fit:
        for i in range(n0, n1f & ~((1 << 30) | (1 << 29))):
            want_w[i, ci[rp[i]:rp[i + 1]] - n0] = vv[rp[i]:rp[i + 1]]
    assert (W == want_w).all()
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V, plan=plan, val=val)
    dQ, dK, dV = gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO, plan=plan, val=val)
    (out_inf,) = gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)
    want = oracle_mod.gt_forward(rp, ci, n_(val), n_(Q), n_(K), n_(V))
    wq, wk, wv = oracle_mod.gt_backward(rp, ci, n_(val), n_(Q), n_(K), n_(V), n_(dO))
    _close(out, want, "out")
    assert torch.equal(out_inf, out)                       # the same kernel without the statistics
    _close(dQ, wq, "dQ")
    _close(dK, wk, "dK")
    _close(dV, wv, "dV")
    # the edge-walking kernels (what weighted edges took before) on the same inputs
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out_a, attn = gt.gt_hyper_forward(*args)
    dQ_a, dK_a, dV_a = gt.gt_backward(*args, attn, dO)
    for a, b, what in ((out, out_a, "out"), (dQ, dQ_a, "dQ"), (dK, dK_a, "dK"), (dV, dV_a, "dV")):
        assert torch.allclose(a, b, atol=5e-5, rtol=2e-4), what


@pytest.mark.parametrize("f", [128, 64, 32, 16, 8])
def test_ranked_pair_equals_the_attn_edge_pair(oracle_mod, f):
    """The attn_edge pair with the values in rank order (dfgnn_gt_hyper_fwd_ranked / dfgnn_gt_bwd_ranked; one head): `out` and
    the gradients are BIT-IDENTICAL to the CSR-ordered pair's (same arithmetic, another edge order in between), the values
    are the CSR ones sorted by column within each row, everything agrees with the oracle, and FusedGTFunction_hyper takes
    this pair at one head."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
    from DFGNN.utils import synthetic as S
    g = _geometry_batch(41 + f)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, 1, f, seed=7, device=DEV)
    dO = torch.randn(m, 1, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    plan = gt.gt_ranked_pair_chosen(row_ptr, col_ind, val, Q)     # (the batch holds ranges of more than 128 nodes)
    assert plan is not None and gt.gt_stats_pair_chosen(row_ptr, col_ind, val, Q) is None
    out, attn_r = gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V, plan=plan)
    dQ, dK, dV = gt.gt_backward_ranked(row_ptr, col_ind, Q, K, V, attn_r, dO, plan=plan)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out_a, attn = gt.gt_hyper_forward(*args)
    dQ_a, dK_a, dV_a = gt.gt_backward(*args, attn, dO)
    assert torch.equal(out, out_a)
    for a, b, what in ((dQ, dQ_a, "dQ"), (dK, dK_a, "dK"), (dV, dV_a, "dV")):
        assert torch.equal(a, b), what
    # rank order: row i's values sorted by column (the synthetic graphs' CSR rows are NOT column-sorted)
    rp, ci = row_ptr.cpu().numpy(), col_ind.cpu().numpy().astype(np.int64)
    row_of = np.repeat(np.arange(m), np.diff(rp))
    order = np.lexsort((ci, row_of))                       # by row, then column
    assert (order != np.arange(nnz)).any()
    assert np.array_equal(attn_r.cpu().numpy().reshape(-1), attn.cpu().numpy().reshape(-1)[order])
    n_ = lambda t: t.detach().cpu().numpy()  # noqa: E731
    want = oracle_mod.gt_forward(rp, n_(col_ind), n_(val), n_(Q), n_(K), n_(V))
    wq, wk, wv = oracle_mod.gt_backward(rp, n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    _close(out, want, "out"); _close(dQ, wq, "dQ"); _close(dK, wk, "dK"); _close(dV, wv, "dV")
    # the autograd function: same tensors through GTConvFuse_hyper
    Qg, Kg, Vg = (t.clone().requires_grad_(True) for t in (Q, K, V))
    o = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Qg, Kg, Vg)
    o.backward(dO)
    assert torch.equal(o.detach(), out) and torch.equal(Qg.grad, dQ) and torch.equal(Kg.grad, dK) and torch.equal(Vg.grad, dV)
    # edge values: not this pair's (they go to the statistics pair)
    assert gt.gt_ranked_pair_applies(row_ptr, col_ind, torch.rand_like(val) + 0.5, Q) is None
    # a batch without ranges of more than 128 nodes: the 256-thread form of the forward (widths 64 and 128)
    gs = S.pattern_like(batch_size=24, seed=3 + f, mean_nodes=90.0, std_nodes=20.0, lo=20, hi=128, mean_deg=30.0).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(gs)
    Q, K, V = S.gt_features(gs.num_nodes(), 1, f, seed=8, device=DEV)
    dO = torch.randn(gs.num_nodes(), 1, f, generator=torch.Generator().manual_seed(5)).to(DEV)
    plan = gt.gt_ranked_pair_chosen(row_ptr, col_ind, val, Q)
    assert plan is not None and plan.num_dense_wide == 0
    out, attn_r = gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V, plan=plan)
    dQ, dK, dV = gt.gt_backward_ranked(row_ptr, col_ind, Q, K, V, attn_r, dO, plan=plan)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out_a, attn = gt.gt_hyper_forward(*args)
    dQ_a, dK_a, dV_a = gt.gt_backward(*args, attn, dO)
    assert torch.equal(out, out_a) and torch.equal(dQ, dQ_a) and torch.equal(dK, dK_a) and torch.equal(dV, dV_a)
    rp, ci = row_ptr.cpu().numpy(), col_ind.cpu().numpy().astype(np.int64)
    order = np.lexsort((ci, np.repeat(np.arange(gs.num_nodes()), np.diff(rp))))
    assert np.array_equal(attn_r.cpu().numpy().reshape(-1), attn.cpu().numpy().reshape(-1)[order])


@pytest.mark.parametrize("h,f", [(2, 64), (4, 32), (8, 16), (3, 32), (2, 128), (2, 8)])
def test_ranked_pair_several_heads(oracle_mod, h, f):
    """The rank-ordered pair with several heads (every head of a range in one workgroup in the forward): bit-identical to
    the CSR-ordered pair, values sorted by column within each row per head."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = _geometry_batch(43 + f)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, h, f, seed=7, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    plan = gt.gt_ranked_pair_applies(row_ptr, col_ind, val, Q)
    assert plan is not None
    out, attn_r = gt.gt_hyper_forward_ranked(row_ptr, col_ind, Q, K, V, plan=plan)
    dQ, dK, dV = gt.gt_backward_ranked(row_ptr, col_ind, Q, K, V, attn_r, dO, plan=plan)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out_a, attn = gt.gt_hyper_forward(*args)
    dQ_a, dK_a, dV_a = gt.gt_backward(*args, attn, dO)
    assert torch.equal(out, out_a)
    for a, b, what in ((dQ, dQ_a, "dQ"), (dK, dK_a, "dK"), (dV, dV_a, "dV")):
        assert torch.equal(a, b), what
    rp, ci = row_ptr.cpu().numpy(), col_ind.cpu().numpy().astype(np.int64)
    order = np.lexsort((ci, np.repeat(np.arange(m), np.diff(rp))))
    assert np.array_equal(attn_r.cpu().numpy(), attn.cpu().numpy()[:, order])
    n_ = lambda t: t.detach().cpu().numpy()  # noqa: E731
    wq, wk, wv = oracle_mod.gt_backward(rp, n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    _close(out, oracle_mod.gt_forward(rp, n_(col_ind), n_(val), n_(Q), n_(K), n_(V)), "out")
    _close(dQ, wq, "dQ"); _close(dK, wk, "dK"); _close(dV, wv, "dV")


def test_plan_edge_bitmaps_equal_the_adjacency():
    """plan.hip: bit c of mask[8 i ..] <=> edge (i, n0 + c); bit r of maskT[8 j ..] <=> edge (n0 + r, j) -- for every
    node of every dense range, including the directed graphs (mask != maskT) and the isolated nodes (all zero)."""
    import ctypes
    import dfgnn_native
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = _geometry_batch(5)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, 1, 32, seed=5, device=DEV)
    gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)   # builds the plan
    plan = row_ptr._dfgnn_plans[32]
    buf = plan.buf.cpu().numpy()
    assert plan.num_dense == plan.num_fit and plan.num_spill == 0
    coords_off = plan.meta[11]
    mask_off = (coords_off + (nnz + 1) // 2 + 4 + 3) & ~3              # dfgnn_launch.hpp: plan_mask_off
    assert mask_off + 16 * m + (nnz + 1) // 2 + 4 == dfgnn_native.lib().dfgnn_plan_ints(m, nnz)   # bitmaps, ranked coordinates
    mask = buf[mask_off:mask_off + 8 * m].view(np.uint32).reshape(m, 8)
    maskT = buf[mask_off + 8 * m:mask_off + 16 * m].view(np.uint32).reshape(m, 8)
    fit = buf[12:12 + 2 * plan.num_fit].reshape(-1, 2)
    rp, ci = row_ptr.cpu().numpy(), col_ind.cpu().numpy()
    want = np.zeros((m, 8), np.uint32)
    wantT = np.zeros((m, 8), np.uint32)
    for n0, n1f in fit:
        n1 = n1f & ~((1 << 30) | (1 << 29))
        for i in range(n0, n1):
            for j in ci[rp[i]:rp[i + 1]]:
                want[i, (j - n0) >> 5] |= np.uint32(1) << np.uint32((j - n0) & 31)
                wantT[j, (i - n0) >> 5] |= np.uint32(1) << np.uint32((i - n0) & 31)
    assert (mask == want).all() and (maskT == wantT).all()
    assert (mask != maskT).any()                                         # (the directed graphs)


def test_autograd_function_takes_the_stats_pair_when_it_applies(oracle_mod, monkeypatch):
    """FusedGTFunction_hyper: the statistics pair on an all-dense batch with several heads or with edge values other than
    ones, the attn_edge pair when one range is not dense (a duplicate edge), at one head with unit values (unless
    DFGNN_STATS=1) and under DFGNN_STATS=0 -- with the same gradients every time."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.operators.fused_gtconv import FusedGTFunction_hyper, GTConvFuse_hyper
    from DFGNN.utils import synthetic as S
    taken = []
    orig = gt.gt_hyper_forward_stats
    monkeypatch.setattr(gt, "gt_hyper_forward_stats", lambda *a, **k: (taken.append("stats"), orig(*a, **k))[1])

    def run(g, weighted=False, heads=2):
        A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
        if weighted:
            val = (torch.rand_like(val) + 0.5)
        m = g.num_nodes()
        Q, K, V = (t.requires_grad_(True) for t in S.gt_features(m, heads, 32, seed=9, device=DEV))
        dO = torch.randn(m, heads, 32, generator=torch.Generator().manual_seed(4)).to(DEV)
        out = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        out.backward(dO)
        n_ = lambda t: t.detach().cpu().numpy()  # noqa: E731
        wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
        _close(out, oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V)), "out")
        _close(Q.grad, wq, "dQ"); _close(K.grad, wk, "dK"); _close(V.grad, wv, "dV")

    run(_geometry_batch(1))
    assert taken == ["stats"]
    run(_geometry_batch(1, duplicate=True))          # one range is not dense: the whole batch takes the attn_edge pair
    assert taken == ["stats"]
    run(_geometry_batch(1), weighted=True)            # edge values: the statistics pair reads them in dense form
    assert taken == ["stats", "stats"]
    run(_geometry_batch(1), weighted=True, heads=1)   # ... at any head count
    assert taken == ["stats"] * 3
    run(_geometry_batch(1, duplicate=True), weighted=True)   # not all dense: the edge-walking kernels
    assert taken == ["stats"] * 3
    run(_geometry_batch(1), heads=1)                  # one head: the attn_edge pair (in rank order) is the faster one
    assert taken == ["stats"] * 3
    monkeypatch.setattr(gt, "STATS_PAIR_MIN_HEADS", 1)  # (DFGNN_STATS=1)
    monkeypatch.setattr(gt, "_STATS_ENV", "1")
    run(_geometry_batch(1), heads=1)
    assert taken == ["stats"] * 4
    monkeypatch.setattr(gt, "USE_STATS_PAIR", False)
    run(_geometry_batch(1))
    run(_geometry_batch(1), weighted=True)
    assert taken == ["stats"] * 4
    assert FusedGTFunction_hyper is not None


@pytest.mark.parametrize("scale_qk,scale_v,scale_do", [(3.0, 1.0, 1.0), (1.0, 1e-6, 1e5), (0.05, 3e4, 1e-7)])
@pytest.mark.parametrize("h,f", [(1, 128), (8, 16)])
def test_stats_pair_is_fp32_equivalent(oracle_mod, scale_qk, scale_v, scale_do, h, f):
    """As test_dense_kernels_are_fp32_equivalent for the pair that recomputes P: un-normalised features (logits of
    +-100: the recomputed exponentials must reproduce the forward's), operands far outside the fp16 range, rows of very
    different magnitude.  Error relative to the largest element of each result <= 2e-4, and no worse than 4x what the
    fp32 edge-walking kernels reach."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import Graph, batch
    rng = np.random.default_rng(23)
    graphs = []
    for n, p in ((40, 0.5), (96, 0.4), (128, 0.4), (131, 0.4), (160, 0.3), (170, 0.3), (255, 0.15)):
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(len(iu)) < p
        graphs.append(Graph(np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]]), n))
    g = batch(graphs).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    gen = torch.Generator().manual_seed(7)
    Q = (torch.randn(m, h, f, generator=gen) * scale_qk * (128 / (h * f)) ** 0.0).to(DEV)
    K = (torch.randn(m, h, f, generator=gen) * scale_qk).to(DEV)
    V = torch.randn(m, h, f, generator=gen) * scale_v
    V[::3] *= 1e-3
    V = V.to(DEV)
    dO = (torch.randn(m, h, f, generator=gen) * scale_do).to(DEV)
    assert gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q)
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    dQ, dK, dV = gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    gt.USE_BLOCK_PLAN = False
    try:
        out_v, attn_v = gt.gt_hyper_forward(*args)
        dQ_v, dK_v, dV_v = gt.gt_backward(*args, attn_v, dO)
    finally:
        gt.USE_BLOCK_PLAN = True
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V))
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))

    def rel(got, ref):
        return float(np.abs(n_(got).astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-300))

    for what, got, valu, ref in (("out", out, out_v, want), ("dQ", dQ, dQ_v, wq), ("dK", dK, dK_v, wk), ("dV", dV, dV_v, wv)):
        assert torch.isfinite(got).all(), what
        e_mc, e_valu = rel(got, ref), rel(valu, ref)
        assert e_mc <= 2e-4, f"{what}: matrix-core error {e_mc:.2e} of the largest element"
        assert e_mc <= 4 * e_valu + 1e-6, f"{what}: matrix-core error {e_mc:.2e} vs fp32 VALU kernels {e_valu:.2e}"


@pytest.mark.parametrize("heads", [1, 8])
def test_stats_pair_c3_full_size_against_the_attn_pair_and_oracle_slices(oracle_mod, heads):
    """BASELINE.json configs[2] at full size (bs = 1024): the pair against the attn_edge pair on every element, and
    against the oracle on the first and last 24 graphs (what the oracle finishes in seconds)."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    f = 128 // heads
    g_host = S.pattern_like(batch_size=1024, seed=1)
    g = g_host.to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, heads, f, seed=100, device=DEV)
    dO = torch.randn(m, heads, f, generator=torch.Generator().manual_seed(7)).to(DEV)
    assert gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q)
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    dQ, dK, dV = gt.gt_backward_stats(row_ptr, col_ind, Q, K, V, mx, sm, dO)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out_a, attn = gt.gt_hyper_forward(*args)
    dQ_a, dK_a, dV_a = gt.gt_backward(*args, attn, dO)
    for a, b, what in ((out, out_a, "out"), (dQ, dQ_a, "dQ"), (dK, dK_a, "dK"), (dV, dV_a, "dV")):
        assert torch.allclose(a, b, atol=2e-5, rtol=1e-4), what
    sizes = g_host.batch_num_nodes().numpy()
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    rp, ci, vl = n_(row_ptr), n_(col_ind), n_(val)
    for lo, hi in ((0, int(sizes[:24].sum())), (m - int(sizes[-24:].sum()), m)):
        e0, e1 = rp[lo], rp[hi]
        srp, sci = rp[lo:hi + 1] - e0, ci[e0:e1] - lo
        sl = slice(lo, hi)
        want = oracle_mod.gt_forward(srp, sci, vl[e0:e1], n_(Q)[sl], n_(K)[sl], n_(V)[sl])
        wq, wk, wv = oracle_mod.gt_backward(srp, sci, vl[e0:e1], n_(Q)[sl], n_(K)[sl], n_(V)[sl], n_(dO)[sl])
        _close(out[sl], want, "out")
        _close(dQ[sl], wq, "dQ"); _close(dK[sl], wk, "dK"); _close(dV[sl], wv, "dV")


@pytest.mark.parametrize("heads", [1, 8])
def test_stats_kernels_write_nothing_outside_their_outputs(heads):
    """Guard regions around every output of the two C-ABI calls (raw pointers, preallocated buffers) stay untouched."""
    import dfgnn_native
    import fused_gtconv as gt
    from _binding_util import get_plan
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    f = 128 // heads
    g = _geometry_batch(3)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, heads, f, seed=5, device=DEV)
    dO = torch.randn_like(Q)
    plan, meta, _ = get_plan(row_ptr, col_ind, f, True)
    G = 4096

    def guarded(numel):
        buf = torch.full((numel + 2 * G,), 12345.0, device=DEV)
        return buf, buf[G:G + numel]

    bufs = {k: guarded(m * heads * f) for k in ("out", "dQ", "dK", "dV")}
    bufs.update({k: guarded(m * heads) for k in ("mx", "sm")})
    L = dfgnn_native.lib()
    P = lambda t: t.data_ptr()  # noqa: E731
    s = torch.cuda.current_stream().cuda_stream
    assert L.dfgnn_gt_hyper_fwd_stats(m, nnz, heads, f, P(row_ptr), P(col_ind), None, P(Q), P(K), P(V), P(bufs["mx"][1]),
                                      P(bufs["sm"][1]), P(bufs["out"][1]), plan, meta, s) == 0
    assert L.dfgnn_gt_bwd_stats(m, nnz, heads, f, P(row_ptr), P(col_ind), None, P(Q), P(K), P(V), P(bufs["mx"][1]), P(bufs["sm"][1]),
                                P(dO), P(bufs["dQ"][1]), P(bufs["dK"][1]), P(bufs["dV"][1]), plan, meta, s) == 0
    torch.cuda.synchronize()
    for k, (buf, view) in bufs.items():
        assert (buf[:G] == 12345.0).all() and (buf[-G:] == 12345.0).all(), k
        assert (view != 12345.0).any(), k
    out, mx, sm = gt.gt_hyper_forward_stats(row_ptr, col_ind, Q, K, V)
    assert torch.equal(out.reshape(-1), bufs["out"][1])
    # argument errors: the pair refuses a batch it does not cover instead of computing something else
    assert L.dfgnn_gt_hyper_fwd_stats(m, nnz, heads, f, P(row_ptr), P(col_ind), None, P(Q), P(K), P(V), P(bufs["mx"][1]),
                                      P(bufs["sm"][1]), P(bufs["out"][1]), None, None, s) == -2
    assert L.dfgnn_gt_bwd_stats(m, nnz, heads, f, P(row_ptr), P(col_ind), None, P(Q), P(K), P(V), None, P(bufs["sm"][1]),
                                P(dO), P(bufs["dQ"][1]), P(bufs["dK"][1]), P(bufs["dV"][1]), plan, meta, s) == -1


def test_operator_runs_in_a_process_that_opened_the_library_before_torch(tmp_path):
    """The C-ABI library opened FIRST, torch imported afterwards, then an operator: one HIP runtime in the process and a
    correct result (csrc/gen_shim.py; round 2: 'invalid argument' on torch's device pointers in exactly this order)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import ctypes, os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, 'df-gnn_amd'), os.path.join(%(root)r, 'tests')]
lib = ctypes.CDLL(os.path.join(%(root)r, 'df-gnn_amd', 'libdfgnn.so'))      # before torch
assert lib.dfgnn_abi_version() == 11 and 'torch' not in sys.modules
import numpy as np, torch
import oracle
from DFGNN.layers import preprocess_Hyper_fw_bw
from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
from DFGNN.utils import synthetic as S
g = S.pattern_like(batch_size=6, seed=2).to('cuda:0')
A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
m = g.num_nodes()
Q, K, V = (t.requires_grad_(True) for t in S.gt_features(m, 1, 128, seed=1, device='cuda:0'))
dO = torch.randn(m, 1, 128, device='cuda:0')
out = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
out.backward(dO)
n = lambda t: t.detach().cpu().numpy()
want = oracle.gt_forward(n(row_ptr), n(col_ind), n(val), n(Q), n(K), n(V))
wq, wk, wv = oracle.gt_backward(n(row_ptr), n(col_ind), n(val), n(Q), n(K), n(V), n(dO))
for got, ref in ((out, want), (Q.grad, wq), (K.grad, wk), (V.grad, wv)):
    assert np.abs(n(got).astype(np.float64) - ref).max() < 1e-3
maps = open('/proc/self/maps').read()
hip = sorted({ln.split()[-1] for ln in maps.splitlines() if 'libamdhip64' in ln})
assert len(hip) == 1, hip                                                      # ONE HIP runtime: torch's
assert 'libdfgnn_hip.so' in maps
print('ok', hip[0])
""" % {"root": ROOT}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]
