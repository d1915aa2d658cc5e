"""GPU parity tests (run with `-m gpu` on an MI355X): every HIP entry point, called through the
Python binding -> C ABI (libdfgnn.so), against the CPU oracle on identical inputs.

Tolerance: BASELINE.json's north_star states "within 1e-3 fp32"; the reference's own check is
isclose(rtol=1e-3) (DFGNN/utils/util.py:211-214).  We assert  |got - want| <= ATOL + RTOL*|want|
with ATOL = RTOL = 1e-3 on O(1)-scaled data, and in practice land near 1e-6.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, csc_of, random_graph

pytestmark = pytest.mark.gpu

ATOL = 1e-3
RTOL = 1e-3
DEV = "cuda:0"


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def _close(got, want, what):
    got = got.detach().cpu().double().numpy()
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = np.abs(got - want)
    bad = err > ATOL + RTOL * np.abs(want)
    assert not bad.any(), f"{what}: {bad.sum()} elements off, max abs err {err.max():.3e}"
    return float(err.max()) if err.size else 0.0


def _loaded_native():
    """The HIP library (not a fallback) is what this process has mapped."""
    maps = open("/proc/self/maps").read()
    assert "libdfgnn.so" in maps


def _gt_all_variants(g, check_attn=True):
    """Run every GT forward entry point on CSR arrays in dict g (numpy) and compare with g['out']."""
    import fused_gtconv as gt
    from DFGNN.operators import fused_gtconv as ops
    ip, idx, rows, val = _t(g["row_ptr"]), _t(g["col_ind"]), _t(g["rows"]), _t(g["val"])
    Q, K, V = _t(g["Q"]), _t(g["K"]), _t(g["V"])
    errs = {}
    errs["hyper"] = _close(ops.GTConvFuse_inference_hyper(ip, idx, rows, val, 1024, Q, K, V), g["out"], "hyper")
    errs["tiling"] = _close(ops.GTConvFuse_inference_tiling(ip, idx, val, 128, Q, K, V), g["out"], "tiling")
    errs["softmax"] = _close(ops.GTConvFuse_inference_softmax(ip, idx, rows, val, 128, Q, K, V), g["out"], "softmax")
    errs["softmax_gm"] = _close(ops.GTConvFuse_inference_softmax_gm(ip, idx, rows, val, Q, K, V), g["out"], "gm")
    errs["csr"] = _close(ops.GTConvFuse_inference_csr(ip, idx, val, 128, Q, K, V), g["out"], "csr")
    errs["csr_gm"] = _close(gt.gt_csr_gm_inference(ip, idx, val, Q, K, V)[0], g["out"], "csr_gm")
    cp, ri, vi = _t(g["col_ptr"]), _t(g["row_ind"]), _t(g["val_idx"])
    out, attn = gt.gt_hyper_forward(ip, idx, rows, val, cp, ri, vi, 1024, Q, K, V)
    errs["fwd_train"] = _close(out, g["out"], "hyper_forward out")
    if check_attn:
        errs["attn"] = _close(attn, g["attn"], "attn_edge")
    return errs


def _gt_backward(g):
    from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
    ip, idx, rows, val = _t(g["row_ptr"]), _t(g["col_ind"]), _t(g["rows"]), _t(g["val"])
    cp, ri, vi = _t(g["col_ptr"]), _t(g["row_ind"]), _t(g["val_idx"])
    Q, K, V = (_t(g[k]).requires_grad_(True) for k in ("Q", "K", "V"))
    out = GTConvFuse_hyper(rows, ip, idx, val, cp, ri, vi, 1024, Q, K, V)
    out.backward(_t(g["dO"]))
    return {k: _close(t.grad, g[k], k) for k, t in (("dQ", Q), ("dK", K), ("dV", V))}


def _gat_all_variants(g):
    from DFGNN.operators import fused_gatconv as ops
    ip, idx, rows = _t(g["row_ptr"]), _t(g["col_ind"]), _t(g["rows"])
    ar, ac, X = _t(g["attn_row"]), _t(g["attn_col"]), _t(g["V"])
    s = float(g["negative_slope"])
    return dict(
        hyper=_close(ops.GATConvFuse_inference_hyper(1024, ar, ac, ip, idx, rows, s, X), g["gat_out"], "gat hyper"),
        softmax=_close(ops.GATConvFuse_inference_softmax(128, ar, ac, ip, idx, rows, s, X), g["gat_out"], "gat sm"),
        softmax_gm=_close(ops.GATConvFuse_inference_softmax_gm(ar, ac, ip, idx, rows, s, X), g["gat_out"], "gat gm"),
        tiling=_close(ops.GATConvFuse_inference_tiling(ar, ac, ip, idx, s, X), g["gat_out"], "gat tiling"),
        csr=_close(ops.GATConvFuse_inference(ar, ac, ip, idx, s, X), g["gat_out"], "gat csr"))


def test_golden_vectors(golden):
    """Committed fixtures: tiny block-diagonal batch, multi-head with isolated nodes, star graphs with
    degree 200 / 1500 (> 64, > 128, > 1024), duplicate edges + self loops + non-unit val, odd f."""
    for name, g in golden.items():
        errs = _gt_all_variants(g)
        errs.update(_gt_backward(g))
        errs.update({"gat_" + k: v for k, v in _gat_all_variants(g).items()})
        print(name, {k: f"{v:.1e}" for k, v in errs.items()})
    _loaded_native()


def _random_case(oracle_mod, seed, m, avg, h, f, unit_val=False, **kw):
    rng = np.random.default_rng(seed)
    indptr, indices, rows = random_graph(rng, m, avg, **kw)
    nnz = len(indices)
    val = np.ones(nnz, np.float32) if unit_val else rng.uniform(0.5, 1.5, nnz).astype(np.float32)
    sc = f ** -0.25
    Q, K, V = (rng.standard_normal((m, h, f)).astype(np.float32) * sc for _ in range(3))
    dO = rng.standard_normal((m, h, f)).astype(np.float32)
    ar, ac = (rng.standard_normal((m, h)).astype(np.float32) for _ in range(2))
    col_ptr, row_ind, val_idx = csc_of(indptr, indices, rows, m)
    g = dict(row_ptr=indptr, col_ind=indices, rows=rows, val=val, col_ptr=col_ptr, row_ind=row_ind,
             val_idx=val_idx, Q=Q, K=K, V=V, dO=dO, attn_row=ar, attn_col=ac, negative_slope=np.float32(0.2))
    g["out"], g["attn"] = oracle_mod.gt_forward(indptr, indices, val, Q, K, V, want_attn=True)
    g["dQ"], g["dK"], g["dV"] = oracle_mod.gt_backward(indptr, indices, val, Q, K, V, dO)
    g["gat_out"] = oracle_mod.gat_forward(indptr, indices, ar, ac, 0.2, V)
    return g


@pytest.mark.parametrize("m,avg,h,f,kw", [
    (1, 0, 1, 4, {}),                                    # one empty row, nnz == 0
    (5, 0, 2, 8, {}),                                    # no edges at all
    (333, 9, 1, 128, dict(empty_frac=0.1)),              # headline width, isolated nodes
    (257, 40, 1, 64, dict(dup_frac=0.3)),                # duplicates
    (64, 4, 8, 16, dict(empty_frac=0.5)),                # many heads, mostly empty
    (100, 30, 4, 32, {}),
    (40, 6, 2, 256, {}),                                 # G = 64 lanes
    (24, 5, 1, 512, {}),                                 # two chunks per lane
    (24, 5, 1, 1024, {}),                                # four chunks per lane
    (50, 7, 3, 20, {}),                                  # f % 4 == 0 but not a power of two
    (50, 7, 2, 7, {}),                                   # scalar path
    (30, 6, 1, 100, {}),                                 # f = 100 (vec4, masked lanes)
    (30, 6, 2, 130, {}),                                 # f % 4 != 0, > 128
    (300, 12, 1, 128, dict(max_deg=5000)),               # one super-node row beyond the LDS budget
    (40, 300, 1, 32, {}),                                # every workgroup beyond the LDS budget (16*300 > 4096)
])
def test_random_graphs_all_entry_points(oracle_mod, m, avg, h, f, kw):
    g = _random_case(oracle_mod, 7 * m + f, m, avg, h, f, **kw)
    _gt_all_variants(g)
    _gt_backward(g)
    _gat_all_variants(g)


def test_pattern_like_batch_c3_shape(oracle_mod):
    """Config 3 at reduced batch (64 graphs, f = 128, h = 1): the shape the headline bench runs."""
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = S.pattern_like(batch_size=64, seed=1).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, 1, 128, seed=1, device=DEV)
    dO = torch.randn(m, 1, 128, generator=torch.Generator().manual_seed(2)).to(DEV)
    case = dict(row_ptr=row_ptr.cpu().numpy(), col_ind=col_ind.cpu().numpy(), rows=rows.cpu().numpy(),
                val=val.cpu().numpy(), col_ptr=col_ptr.cpu().numpy(), row_ind=row_ind.cpu().numpy(),
                val_idx=val_idx.cpu().numpy(), Q=Q.cpu().numpy(), K=K.cpu().numpy(), V=V.cpu().numpy(),
                dO=dO.cpu().numpy())
    case["out"], case["attn"] = oracle_mod.gt_forward(case["row_ptr"], case["col_ind"], case["val"], case["Q"],
                                                      case["K"], case["V"], want_attn=True)
    case["dQ"], case["dK"], case["dV"] = oracle_mod.gt_backward(case["row_ptr"], case["col_ind"], case["val"],
                                                                case["Q"], case["K"], case["V"], case["dO"])
    print(_gt_all_variants(case), _gt_backward(case))


def test_cora_and_reddit_like_gat(oracle_mod):
    """Configs 2 and 4 (reddit at 1% scale so the oracle finishes in seconds)."""
    from DFGNN.layers.util import preprocess_CSR, preprocess_softmax
    from DFGNN.operators import fused_gatconv as ops
    from DFGNN.utils import synthetic as S
    for graph, f in ((S.cora_like(), 128), (S.reddit_like(scale=0.01), 128)):
        g = graph.to(DEV)
        row_ptr, col_ind, rows, _, smem = preprocess_softmax(g)
        m = g.num_nodes()
        ar, ac, X = S.gat_features(m, 1, f, seed=4, device=DEV)
        want = oracle_mod.gat_forward(row_ptr.cpu().numpy(), col_ind.cpu().numpy(), ar.cpu().numpy(),
                                      ac.cpu().numpy(), 0.2, X.cpu().numpy())
        _close(ops.GATConvFuse_inference_softmax(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X), want, "softmax")
        _close(ops.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X), want, "tiling")
        _close(ops.GATConvFuse_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X), want, "hyper")
        assert len(preprocess_CSR(g)) == 4


def test_full_size_properties_c3():
    """Config 3 at BASELINE.json's full size (bs = 1024): size-independent properties instead of the
    oracle -- attention rows sum to 1, V == 1 gives out == 1 on non-empty rows, the four variants agree
    with each other, and <dO, out> == <dV, V> (linearity of out in V, checks the CSC pass)."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = S.pattern_like(batch_size=1024, seed=1).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = S.gt_features(m, 1, 128, seed=1, device=DEV)
    out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    rowsum = torch.zeros(m, device=DEV).index_add_(0, rows.long(), attn[0])
    deg = (row_ptr[1:] - row_ptr[:-1])
    assert torch.allclose(rowsum, (deg > 0).float(), atol=1e-4)
    ones = torch.ones_like(V)
    o1 = gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, ones)[0]
    assert torch.allclose(o1[deg > 0], ones[deg > 0], atol=1e-4)
    for other in (gt.gt_tiling_inference(row_ptr, col_ind, val, 128, Q, K, V)[0],
                  gt.gt_softmax_inference(row_ptr, col_ind, rows, val, 128, Q, K, V)[0],
                  gt.gt_softmax_gm_inference(row_ptr, col_ind, rows, val, Q, K, V)):
        assert torch.allclose(other, out, atol=1e-4, rtol=1e-3)
    dO = torch.randn_like(out)
    dQ, dK, dV = gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO)
    lhs = (dO.double() * out.double()).sum()
    rhs = (dV.double() * V.double()).sum()
    # both sides are cancellation-heavy sums of ~1.8e7 terms: the error bar is relative to the sum of magnitudes
    assert abs(float(lhs - rhs)) <= 1e-6 * float((dO.double() * out.double()).abs().sum())
    # softmax shift invariance: dS sums to zero per row  =>  <dQ, Q> == <dK, K>
    a = (dQ.double() * Q.double()).sum()
    b = (dK.double() * K.double()).sum()
    assert abs(float(a - b)) <= 1e-5 * float((dQ.double() * Q.double()).abs().sum())


@pytest.mark.parametrize("heads", [4, 8])
def test_c5_peptides_like_full_size_against_oracle(oracle_mod, heads):
    """BASELINE.json configs[4] at its full size: GT conv on a Peptides-struct-like batch (bs = 256, dim = 128,
    heads 4 / 8, 'hyper' fwd + bwd training step; low-degree lane-group kernels, isolated nodes) and the GAT training
    pair on the same batch, all against the oracle."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.operators.fused_gatconv import GATConvFuse
    from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
    from DFGNN.utils import synthetic as S
    g = S.peptides_like(batch_size=256, seed=3).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, f = g.num_nodes(), 128 // heads
    assert int((row_ptr[1:] == row_ptr[:-1]).sum()) >= 1            # isolated nodes exist
    Q, K, V = (t.requires_grad_(True) for t in S.gt_features(m, heads, f, seed=3, device=DEV))
    dO = torch.randn(m, heads, f, generator=torch.Generator().manual_seed(5)).to(DEV)
    out = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    dQ, dK, dV = torch.autograd.grad(out, (Q, K, V), dO)
    n_ = lambda t: t.detach().cpu().numpy()  # noqa: E731
    want, want_attn = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), want_attn=True)
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    for got, ref, what in ((out, want, "out"), (dQ, wq, "dQ"), (dK, wk, "dK"), (dV, wv, "dV")):
        _close(got, ref, f"C5 heads={heads} {what}")
    _, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q.detach(), K.detach(),
                                  V.detach())
    _close(attn, want_attn, f"C5 heads={heads} attn_edge")
    _close(gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q.detach(), K.detach(), V.detach())[0], want,
           f"C5 heads={heads} inference")
    ar, ac, X = (t.requires_grad_(True) for t in S.gat_features(m, heads, f, seed=9, device=DEV))
    o = GATConvFuse(ar, ac, row_ptr, col_ind, col_ptr, row_ind, val_idx, 0.2, X, 0.0)
    gx, gr, gc = torch.autograd.grad(o, (X, ar, ac), dO)
    args = (n_(row_ptr), n_(col_ind), n_(ar), n_(ac), 0.2, n_(X))
    w_out, _, _ = oracle_mod.gat_train_forward(*args, None, 0.0)
    w_gf, w_gr, w_gc = oracle_mod.gat_backward(*args, n_(dO), None, 0.0)
    for got, ref, what in ((o, w_out, "out"), (gx, w_gf, "grad_feat"), (gr, w_gr, "grad_attn_row"), (gc, w_gc, "grad_attn_col")):
        _close(got, ref, f"C5 heads={heads} GAT training {what}")


@pytest.mark.parametrize("heads", [1, 8])
def test_c3_full_size_slices_against_oracle(oracle_mod, heads):
    """BASELINE.json configs[2] at its full size (bs = 1024, dim = 128, 'hyper' fwd + bwd on the matrix-core kernels):
    the batch is block-diagonal, so the oracle applied to a run of whole graphs equals that slice of the full result.
    The first 128 and the last 128 graphs (a quarter of the batch) are checked: out, attn_edge, dQ, dK, dV.
    heads = 8 (dim 128 as 8 heads of 16, `bench.py --heads 8`): the same batch through the multi-head bodies -- one
    workgroup per graph takes the heads (dfgnn_dense_heads.hpp), the 1-D backward grid with per-head and walking
    workgroups side by side."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g_host = S.pattern_like(batch_size=1024, seed=1)
    g = g_host.to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    f = 128 // heads
    Q, K, V = S.gt_features(m, heads, f, seed=100, device=DEV)
    dO = torch.randn(m, heads, f, generator=torch.Generator().manual_seed(7)).to(DEV)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    plan = row_ptr._dfgnn_plans[f]
    assert plan.num_dense == plan.num_fit > 1000 and plan.num_spill == 0
    dQ, dK, dV = gt.gt_backward(*args, attn, dO)
    sizes = g_host.batch_num_nodes().numpy()
    off = np.concatenate([[0], np.cumsum(sizes)])
    rp = row_ptr.cpu().numpy()
    n_ = lambda t: t.detach().cpu().numpy()  # noqa: E731
    for g0, g1 in ((0, 128), (896, 1024)):
        n0, n1 = int(off[g0]), int(off[g1])
        e0, e1 = int(rp[n0]), int(rp[n1])
        srp = rp[n0:n1 + 1] - e0
        sci = n_(col_ind[e0:e1]) - n0
        sv = n_(val[e0:e1])
        q, k, v, do = (n_(t[n0:n1]) for t in (Q, K, V, dO))
        want, want_attn = oracle_mod.gt_forward(srp, sci, sv, q, k, v, want_attn=True)
        wq, wk, wv = oracle_mod.gt_backward(srp, sci, sv, q, k, v, do)
        _close(out[n0:n1], want, f"C3 heads={heads} graphs {g0}:{g1} out")
        _close(attn[:, e0:e1], want_attn, f"C3 heads={heads} graphs {g0}:{g1} attn_edge")
        for got, ref, what in ((dQ, wq, "dQ"), (dK, wk, "dK"), (dV, wv, "dV")):
            _close(got[n0:n1], ref, f"C3 heads={heads} graphs {g0}:{g1} {what}")


def test_c4_reddit_like_full_size_spot_rows():
    """BASELINE.json configs[3] at its full size (232 965 nodes, 114.6 M edges, max degree ~20 k, GAT 'tiling', f = 128):
    rows are independent, so a sample of rows -- the 20 heaviest, 20 empty / lightest and 200 random ones -- is
    recomputed in float64 numpy from the same CSR arrays (a full oracle pass over 114 M edges is a benchmark, not a
    test), plus size-independent properties: X == 1 gives out == 1 on every non-empty row and 0 on empty ones."""
    from DFGNN.layers.util import preprocess_CSR
    from DFGNN.operators import fused_gatconv as ops
    from DFGNN.utils import synthetic as S
    g = S.reddit_like().to(DEV)
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    m, nnz = g.num_nodes(), g.num_edges()
    assert m == 232965 and nnz > 110_000_000
    ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=DEV)
    out = ops.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)
    deg = (row_ptr[1:] - row_ptr[:-1]).cpu()
    order = torch.argsort(deg)
    pick = torch.cat([order[-20:], order[:20], torch.randperm(m, generator=torch.Generator().manual_seed(0))[:200]]).unique()
    cp = row_ptr.cpu().numpy()
    worst = 0.0
    for r in pick.numpy():
        cols = col_ind[cp[r]:cp[r + 1]].long()
        ref = np.zeros(128)
        if len(cols):
            sc = ar[r, 0].double().item() + ac[cols, 0].double().cpu().numpy()
            sc = np.where(sc > 0, sc, 0.2 * sc)
            p = np.exp(sc - sc.max())
            ref = (p[:, None] * X[cols, 0].double().cpu().numpy()).sum(0) / p.sum()
        err = np.abs(out[r, 0].double().cpu().numpy() - ref)
        assert (err <= ATOL + RTOL * np.abs(ref)).all(), f"row {r} (degree {int(deg[r])}): max abs err {err.max():.3e}"
        worst = max(worst, float(err.max()))
    assert int(deg.max()) > 15000 and worst < 1e-4
    # every row of the output against the oracle's fp32 build on the host cores (a pass over 114.6 M edges: seconds), and
    # the chunked form (what the operator runs at this size: fused_gatconv._use_chunked_tiling) against the single kernel
    import fused_gatconv as gat
    import oracle
    assert gat._use_chunked_tiling(m, nnz, 1, 128)
    full = oracle.gat_forward(row_ptr.cpu().numpy(), col_ind.cpu().numpy(), ar.cpu().numpy(), ac.cpu().numpy(), 0.2,
                              X.cpu().numpy(), acc="f32")
    err = np.abs(out.cpu().numpy().astype(np.float64) - full)
    assert (err <= ATOL + RTOL * np.abs(full)).all(), f"full output: max abs err {err.max():.3e}"
    rows_tmp = gat.TILING_CHUNK_ROWS
    try:
        gat.TILING_CHUNK_ROWS = 0
        single = ops.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)
    finally:
        gat.TILING_CHUNK_ROWS = rows_tmp
    assert torch.allclose(out, single, atol=1e-5, rtol=1e-4) and not torch.equal(out, single)   # two code paths did run
    o1 = ops.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, torch.ones_like(X))
    nonempty = (deg > 0).to(DEV)
    assert torch.allclose(o1[nonempty], torch.ones_like(o1[nonempty]), atol=1e-4) and bool((o1[~nonempty] == 0).all())


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_batch_equals_unsharded(world):
    """Multi-GPU path on one GPU: the shards shard_graph cuts for `world` ranks (whole graphs, edge-balanced), each run
    through the HIP path on its own, concatenated in rank order == the unsharded batch's result (fwd + bwd; a shard
    only changes which ranges the plan merges, i.e. the power-of-two operand scales: equal to fp32 rounding)."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.parallel import shard_graph
    from DFGNN.utils import synthetic as S
    g_host = S.pattern_like(batch_size=96, seed=21)
    m = g_host.num_nodes()
    Q, K, V = S.gt_features(m, 1, 128, seed=3, device=DEV)
    dO = torch.randn(m, 1, 128, generator=torch.Generator().manual_seed(4)).to(DEV)

    def run(graph, sl):
        A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(graph.to(DEV))
        a = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q[sl].contiguous(), K[sl].contiguous(),
             V[sl].contiguous())
        out, attn = gt.gt_hyper_forward(*a)
        return [out] + list(gt.gt_backward(*a, attn, dO[sl].contiguous()))

    full = run(g_host, slice(0, m))
    parts, covered = [], 0
    for rank in range(world):
        sub, (n0, n1) = shard_graph(g_host, rank, world)
        assert n0 == covered and sub.num_nodes() == n1 - n0
        covered = n1
        parts.append(run(sub, slice(n0, n1)))
    assert covered == m
    for k, what in enumerate(("out", "dQ", "dK", "dV")):
        got = torch.cat([p[k] for p in parts])
        assert got.shape == full[k].shape
        assert torch.allclose(got, full[k], atol=2e-6, rtol=1e-5), (what, float((got - full[k]).abs().max()))


def test_layers_fused_vs_baseline():
    """Layer level, the reference's own check: same module, fuse=False vs fuse=True, check_correct
    (DFGNN/utils/util.py:211-236) on the first / last 1000 rows."""
    import argparse
    from DFGNN.layers import load_graphconv_layer, load_prepfunc
    from DFGNN.utils import check_correct, preprocess_dglsp
    from DFGNN.utils import synthetic as S
    torch.manual_seed(0)
    g = S.pattern_like(batch_size=16, seed=3).to(DEV)
    x = torch.randn(g.num_nodes(), 64, device=DEV)
    for conv in ("gt", "gat", "agnn", "dotgat"):
        for fmt in ("hyper", "softmax", "softmax_gm", "tiling", "csr", "csr_gm", "hyper_ablation", "hyper_v2",
                    "hyper_recompute"):
            if conv == "dotgat" and fmt not in ("hyper", "csr", "softmax"):
                continue                                    # the reference's DOTGAT layer classes (layers/GAT_DOT)
            if conv != "gat" and fmt in ("hyper_v2", "hyper_recompute"):
                continue                                    # GAT-only variants (reference layers/util.py:403-406)
            if (conv == "agnn" and fmt == "hyper_ablation") or (conv == "gat" and fmt == "csr_gm"):
                continue                                    # not in the reference's tables (:424-442, :392-420)
            args = argparse.Namespace(conv=conv, format=fmt, dim=64, heads=1)
            torch.manual_seed(sum(map(ord, conv + fmt)))    # weights do not depend on the order of this loop
            layer = load_graphconv_layer(args).to(DEV).eval()
            with torch.no_grad():
                if conv == "dotgat":                        # params carry the graph itself for the non-fused branch
                    params = load_prepfunc(args)(g)
                    base, _ = layer(params, x * 0.2, fuse=False)
                    fused, ms = layer(params, x * 0.2, fuse=True)
                else:
                    base, _ = layer(preprocess_dglsp(g), x, fuse=False)
                    fused, ms = layer(load_prepfunc(args)(g), x, fuse=True)
            assert torch.allclose(base, fused, atol=1e-4, rtol=1e-3), (conv, fmt)
            ok = check_correct(base[:1000], fused[:1000]) and check_correct(base[-1000:], fused[-1000:])
            assert ok, (conv, fmt)
            assert ms > 0


def test_training_layer_grads_match_autograd_baseline():
    """check_grad of the reference's trainer (train_batch_graph_timing.py:75-112): q/k/v_proj weight
    grads of the fused path vs the non-fused torch path (heads = 1, where the two layouts coincide)."""
    from DFGNN.layers import SparseMHA_forward, preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    torch.manual_seed(1)
    g = S.pattern_like(batch_size=8, seed=4).to(DEV)
    params = preprocess_Hyper_fw_bw(g)
    layer = SparseMHA_forward(64, 64, 1).to(DEV).train()
    x = torch.randn(g.num_nodes(), 64, device=DEV)
    grads = []
    for fuse in (False, True):
        layer.zero_grad()
        out = layer(params, x, fuse=fuse)
        (out * torch.linspace(-1, 1, out.numel(), device=DEV).reshape(out.shape)).sum().backward()
        grads.append([p.grad.clone() for p in (layer.q_proj.weight, layer.k_proj.weight, layer.v_proj.weight)])
    for a, b in zip(*grads):
        assert torch.allclose(a, b, atol=1e-2, rtol=1e-3)   # the reference's own atol
        assert torch.allclose(a, b, atol=2e-4, rtol=1e-3)   # and a tighter one


def test_non_default_stream_and_error_paths():
    import fused_gtconv as gt
    g = torch.Generator().manual_seed(0)
    ip = torch.tensor([0, 2, 3], dtype=torch.int32, device=DEV)
    idx = torch.tensor([0, 1, 1], dtype=torch.int32, device=DEV)
    rows = torch.tensor([0, 0, 1], dtype=torch.int32, device=DEV)
    val = torch.ones(3, device=DEV)
    Q = torch.randn(2, 1, 8, generator=g).to(DEV)
    ref = gt.gt_hyper_inference(ip, idx, rows, val, 1024, Q, Q, Q)[0]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = gt.gt_hyper_inference(ip, idx, rows, val, 1024, Q, Q, Q)[0]
    s.synchronize()
    assert torch.equal(out, ref)
    with pytest.raises(RuntimeError, match="contiguous"):
        gt.gt_hyper_inference(ip, idx, rows, val, 1024, torch.zeros(2, 1, 16, device=DEV)[:, :, ::2], Q, Q)
    with pytest.raises(RuntimeError, match="dtype"):
        gt.gt_hyper_inference(ip.long(), idx, rows, val, 1024, Q, Q, Q)
    with pytest.raises(RuntimeError, match="unsupported"):
        big = torch.zeros(2, 1, 2048, device=DEV)
        gt.gt_hyper_inference(ip, idx, rows, val, 1024, big, big, big)


# ---- block plan (LDS-resident per-graph kernels) ------------------------------------------------
def _natural_ranges(row_ptr, col_ind):
    """Closed contiguous node ranges, computed the slow obvious way on the host."""
    m = len(row_ptr) - 1
    hi = np.arange(m)
    lo = np.arange(m)
    for i in range(m):
        c = col_ind[row_ptr[i]:row_ptr[i + 1]]
        if len(c):
            hi[i] = max(i, c.max())
            lo[i] = min(i, c.min())
    pm = np.maximum.accumulate(hi)
    sm = np.minimum.accumulate(lo[::-1])[::-1]
    ends = [i + 1 for i in range(m) if pm[i] <= i and (i + 1 == m or sm[i + 1] >= i + 1)]
    return ends


def test_block_plan_structure():
    from _binding_util import build_plan
    from DFGNN.layers.util import preprocess_Hyper
    from DFGNN.utils import synthetic as S
    g = S.pattern_like(batch_size=40, seed=9)
    sizes = g.batch_num_nodes().numpy()
    row_ptr, col_ind, rows, val, smem = preprocess_Hyper(g.to(DEV))
    plan = build_plan(row_ptr, col_ind, 128)
    nfit, nspill, maxn, maxe = plan.meta[:4]
    buf = plan.buf.cpu().numpy()
    m = g.num_nodes()
    raw = buf[12:12 + 2 * nfit].reshape(-1, 2)
    FLAGS = (1 << 30) | (1 << 29)               # per-edge array in global scratch | dense (matrix-core kernels)
    fit = raw & ~FLAGS
    rp = row_ptr.cpu().numpy()
    edges_in_order = rp[fit[:, 1]] - rp[fit[:, 0]]
    dense = (raw[:, 1] & (1 << 29)) != 0
    nd = plan.num_dense
    assert dense[:nd].all() and not dense[nd:].any()                                     # dense ranges come first,
    nn = (fit[:, 1] - fit[:, 0]).astype(np.int64)
    nbig = int((nn[:nd] > 128).sum())
    assert (nn[:nbig] > 128).all()                                                       # those of > 128 nodes first,
    for a, b in ((0, nbig), (nbig, nd), (nd, nfit)):                                     # each group largest first
        assert (np.diff(edges_in_order[a:b]) <= 0).all()
    assert (dense == ((nn <= 255) & (32 * edges_in_order >= nn * nn))).all()             # no duplicate edges: all eligible
    assert nd > 0
    fit = fit[np.argsort(fit[:, 0])]
    ends = _natural_ranges(row_ptr.cpu().numpy(), col_ind.cpu().numpy())
    assert nspill == 0 and nfit >= 1
    assert fit[0, 0] == 0 and fit[-1, 1] == m and (fit[1:, 0] == fit[:-1, 1]).all()      # a partition of the rows
    assert set(fit[:, 1]).issubset(set(ends))                                            # cuts only at closed ranges
    assert set(np.cumsum(sizes)).issuperset(set(fit[:, 1]))                              # = graph boundaries
    assert maxn == (fit[:, 1] - fit[:, 0]).max() and maxn <= 256
    assert maxe == (rp[fit[:, 1]] - rp[fit[:, 0]]).max()
    # f = 4: no matrix-core form; many graphs merge into one block of <= 256 nodes
    plan4 = build_plan(row_ptr, col_ind, 4)
    fit4 = plan4.buf.cpu().numpy()[12:12 + 2 * plan4.meta[0]].reshape(-1, 2) & ~FLAGS
    assert plan4.num_dense == 0
    assert plan4.meta[0] < nfit and (fit4[:, 1] - fit4[:, 0]).max() <= 256
    # f = 8 / 16 run zero-padded on the 32-wide matrix-core layout: same classification as f = 128
    for fp in (8, 16):
        planp = build_plan(row_ptr, col_ind, fp)
        assert planp.num_dense == nd and planp.meta[0] == nfit
    # a full graph is one closed range that cannot fit: everything spills in <= 16-row chunks
    c = S.cora_like()
    rp2, ci2, _, _, _ = preprocess_Hyper(c.to(DEV))
    p2 = build_plan(rp2, ci2, 128)
    assert p2.meta[0] == 0 and p2.meta[1] == (c.num_nodes() + 15) // 16


@pytest.mark.parametrize("unit_val", [False, True])
@pytest.mark.parametrize("h,f,bs", [(1, 128, 24), (4, 32, 24), (2, 64, 10), (1, 16, 60), (1, 256, 6), (1, 512, 3)])
def test_block_kernels_match_oracle_and_general_path(oracle_mod, h, f, bs, unit_val):
    """'hyper' with the block plan (K/V resident in LDS) vs the oracle and vs the plan-less kernels.
    unit_val: all-ones edge values reach the C ABI as NULL and take the one-pass forward (K and V resident
    together); weighted edges take the two-pass forward -- or, for inference on an all-dense batch, the matrix-core
    forward with the values in the plan's dense form."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    kw = dict(mean_nodes=40, std_nodes=5, lo=30, hi=50, mean_deg=15) if f >= 512 else {}  # must fit 160 KB of LDS
    g = S.pattern_like(batch_size=bs, seed=21 + f, **kw).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    if not unit_val:
        val = (torch.rand_like(val) + 0.5)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, h, f, seed=3, device=DEV)
    want, want_attn = oracle_mod.gt_forward(row_ptr.cpu().numpy(), col_ind.cpu().numpy(), val.cpu().numpy(),
                                            Q.cpu().numpy(), K.cpu().numpy(), V.cpu().numpy(), want_attn=True)
    try:
        gt.USE_BLOCK_PLAN = True
        out_p = gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0]
        out_t, attn_t = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        assert row_ptr._dfgnn_plans[f].num_fit > 0
        if f in (16, 32, 64, 128):   # widths with a matrix-core form: the ranges are dense, and the inference above ran on
            # the matrix cores with or without edge values (weights in the plan's dense form, csrc/gt_dense_stats_w.hip)
            assert row_ptr._dfgnn_plans[f].num_dense > 0
            assert gt.gt_stats_pair_applies(row_ptr, col_ind, val, Q) is not None
        gt.USE_BLOCK_PLAN = False
        out_g = gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0]
    finally:
        gt.USE_BLOCK_PLAN = True
    _close(out_p, want, "block inference")
    _close(out_t, want, "block training fwd")
    _close(attn_t, want_attn, "block attn_edge")
    assert torch.allclose(out_p, out_g, atol=1e-5, rtol=1e-4)
    # backward through the resident kernel vs the oracle and vs the general two-launch path
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(5)).to(DEV)
    wq, wk, wv = oracle_mod.gt_backward(row_ptr.cpu().numpy(), col_ind.cpu().numpy(), val.cpu().numpy(),
                                        Q.cpu().numpy(), K.cpu().numpy(), V.cpu().numpy(), dO.cpu().numpy())
    dQ, dK, dV = gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn_t, dO)
    _close(dQ, wq, "block dQ")
    _close(dK, wk, "block dK")
    _close(dV, wv, "block dV")
    try:
        gt.USE_BLOCK_PLAN = False
        gQ, gK, gV = gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn_t, dO)
    finally:
        gt.USE_BLOCK_PLAN = True
    for a, b in ((dQ, gQ), (dK, gK), (dV, gV)):
        assert torch.allclose(a, b, atol=1e-4, rtol=1e-3)
    # GAT 'hyper' through the resident kernel (X resident) vs the oracle and the general kernel
    import fused_gatconv as gat
    ar, ac, X = S.gat_features(m, h, f, seed=6, device=DEV)
    want_gat = oracle_mod.gat_forward(row_ptr.cpu().numpy(), col_ind.cpu().numpy(), ar.cpu().numpy(),
                                      ac.cpu().numpy(), 0.2, X.cpu().numpy())
    out_b = gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X)
    try:
        gat.USE_BLOCK_PLAN = False
        out_n = gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X)
    finally:
        gat.USE_BLOCK_PLAN = True
    _close(out_b, want_gat, "block GAT hyper")
    assert torch.allclose(out_b, out_n, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("h,f", [(1, 128), (2, 64), (3, 32), (4, 16), (2, 8)])
def test_dense_kernels_every_geometry(oracle_mod, h, f):
    """Matrix-core kernels (gt_dense.hip) vs the oracle on a batch whose ranges hit every code path: tiny graphs
    (1 strip), <= 128 nodes (one tile), 129-160 nodes (160-row images, two row blocks), 161-255 nodes (2 x 2 tiles),
    a graph with isolated nodes / empty rows, and one graph with a duplicate edge (must NOT be marked dense)."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import Graph, batch
    rng = np.random.default_rng(17 + f)

    def er(n, p, drop_rows=()):
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(len(iu)) < p
        keep &= ~np.isin(iu, drop_rows) & ~np.isin(ju, drop_rows)
        return np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]])

    graphs = []
    for n, p in ((9, 0.9), (17, 0.6), (64, 0.5), (128, 0.35), (129, 0.3), (145, 0.4), (160, 0.3), (161, 0.3), (200, 0.25),
                 (255, 0.2)):
        s_, d_ = er(n, p)
        graphs.append(Graph(s_, d_, n))
    s_, d_ = er(70, 0.5, drop_rows=(0, 33, 69))                     # isolated nodes: empty rows and columns
    graphs.append(Graph(s_, d_, 70))
    keep = rng.random((80, 140)) < 0.3                               # directed: rows 80..139 have in-edges only, so
    ds_, dd_ = np.nonzero(keep)                                       # the second row block of the backward is empty
    graphs.append(Graph(ds_.astype(np.int64), dd_.astype(np.int64), 140))
    s_, d_ = er(40, 0.6)
    graphs.append(Graph(np.concatenate([s_, s_[:1]]), np.concatenate([d_, d_[:1]]), 40))   # one duplicate edge
    g = batch(graphs).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    from DFGNN.utils import synthetic as S
    Q, K, V = S.gt_features(m, h, f, seed=5, device=DEV)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    plan = row_ptr._dfgnn_plans[f]
    raw = plan.buf.cpu().numpy()[12:12 + 2 * plan.num_fit].reshape(-1, 2)
    n0s, n1s = raw[:, 0], raw[:, 1] & ~((1 << 30) | (1 << 29))
    nn = n1s - n0s
    dense = (raw[:, 1] & (1 << 29)) != 0
    assert dense[:plan.num_dense].all() and not dense[plan.num_dense:].any()
    dn = nn[dense]
    assert (dn <= 128).any() and ((dn > 128) & (dn <= 160)).any() and (dn > 160).any() and dn.max() == 255
    dup0 = m - 40                                                    # first node of the graph with the duplicate edge
    hit = (n0s < m) & (n1s > dup0)
    assert hit.any() and not dense[hit].any()                        # its range must stay on the edge-walking kernels
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want, want_attn = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), want_attn=True)
    _close(out, want, "dense fwd")
    _close(attn, want_attn, "dense attn_edge")
    _close(gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0], want, "dense inference")
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    dQ, dK, dV = gt.gt_backward(*args, attn, dO)
    _close(dQ, wq, "dense dQ")
    _close(dK, wk, "dense dK")
    _close(dV, wv, "dense dV")
    # GAT 'hyper' over the same ranges (rank-one logits, P X on the matrix cores)
    import fused_gatconv as gat
    ar, ac, X = S.gat_features(m, h, f, seed=8, device=DEV)
    want_gat = oracle_mod.gat_forward(n_(row_ptr), n_(col_ind), n_(ar), n_(ac), 0.2, n_(X))
    _close(gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X), want_gat, "dense GAT hyper")


def test_gat_forward_tb_returns_the_training_forward():
    """fused_gatconv.gat_forward_tb (fused_gatconv.cpp:256-282): out / edge_max / edge_sum of the GAT forward, whatever
    the tile schedule says about who computes which 32-edge tile."""
    import fused_gatconv as gat
    from DFGNN.utils import synthetic as S
    from DFGNN.layers import preprocess_CSR
    g = S.cora_like().to(DEV)
    row_ptr, col_ind, _, _ = preprocess_CSR(g)
    m = g.num_nodes()
    ar, ac, X = S.gat_features(m, 2, 32, seed=2, device=DEV)
    deg = (row_ptr[1:] - row_ptr[:-1]).cpu().numpy()
    sched = np.array([(r, t) for r in range(m) for t in range((int(deg[r]) + 31) // 32)], dtype=np.int32)
    out, mx, sm = gat.gat_forward_tb(ar, ac, row_ptr, col_ind, 0.2, X, torch.from_numpy(sched).to(DEV))
    ref = gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0)
    assert torch.equal(out, ref[0]) and torch.equal(mx, ref[1]) and torch.equal(sm, ref[2])


def _dense_batch(sizes, seed):
    """Block-diagonal batch of symmetric Erdos-Renyi graphs, given as (nodes, edge probability) pairs (every range
    dense, no duplicates)."""
    from DFGNN.utils import Graph, batch
    rng = np.random.default_rng(seed)
    graphs = []
    for n, p in sizes:
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(len(iu)) < p
        graphs.append(Graph(np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]]), n))
    return batch(graphs).to(DEV)


def _rel_to_max(got, want):
    got = got.detach().cpu().double().numpy()
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-300))


@pytest.mark.parametrize("scale_qk,scale_v,scale_do", [(3.0, 1.0, 1.0),          # un-normalised: |logit| ~ 30-100
                                                        (1.0, 1e-6, 1e5),          # tiny V, huge dO
                                                        (0.05, 3e4, 1e-7)])        # flat softmax, huge V, tiny dO
def test_dense_kernels_are_fp32_equivalent(oracle_mod, scale_qk, scale_v, scale_do):
    """The matrix-core kernels compute with fp16 hi + lo operand halves under power-of-two scales (dfgnn_dense.hpp),
    which must behave like fp32 arithmetic for ANY operand magnitude: un-normalised features (logits of +-100, nearly
    one-hot attention), operands far outside the fp16 range, rows of very different magnitude inside one graph.  The
    bar here is not the 1e-3 parity bar but fp32's own: the error relative to the largest element of each result must
    stay below 2e-4 (logits of magnitude 100 carry ~2e-5 of fp32 rounding themselves), and no worse than 4x what the
    fp32 edge-walking kernels (plan-less path) reach on the same inputs."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    g = _dense_batch(((40, 0.5), (96, 0.4), (128, 0.4), (131, 0.4), (160, 0.3), (170, 0.3), (255, 0.15)), seed=23)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, h, f = g.num_nodes(), 1, 128
    gen = torch.Generator().manual_seed(7)
    Q = (torch.randn(m, h, f, generator=gen) * scale_qk).to(DEV)
    K = (torch.randn(m, h, f, generator=gen) * scale_qk).to(DEV)
    V = (torch.randn(m, h, f, generator=gen) * scale_v)
    V[::3] *= 1e-3                                                   # rows of very different magnitude in one image
    V = V.to(DEV)
    dO = (torch.randn(m, h, f, generator=gen) * scale_do).to(DEV)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    plan = row_ptr._dfgnn_plans[f]
    assert plan.num_dense == plan.num_fit > 0 and plan.num_spill == 0
    dQ, dK, dV = gt.gt_backward(*args, attn, dO)
    gt.USE_BLOCK_PLAN = False
    try:
        out_v, attn_v = gt.gt_hyper_forward(*args)
        dQ_v, dK_v, dV_v = gt.gt_backward(*args, attn_v, dO)
    finally:
        gt.USE_BLOCK_PLAN = True
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want, want_attn = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), want_attn=True)
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    for what, got, valu, ref in (("out", out, out_v, want), ("attn_edge", attn, attn_v, want_attn), ("dQ", dQ, dQ_v, wq),
                                 ("dK", dK, dK_v, wk), ("dV", dV, dV_v, wv)):
        assert torch.isfinite(got).all(), what
        e_mc, e_valu = _rel_to_max(got, ref), _rel_to_max(valu, ref)
        assert e_mc <= 2e-4, f"{what}: matrix-core error {e_mc:.2e} of the largest element"
        assert e_mc <= 4 * e_valu + 1e-6, f"{what}: matrix-core error {e_mc:.2e} vs fp32 VALU kernels {e_valu:.2e}"
    # GAT 'hyper': P X on the matrix cores, X far outside the fp16 range
    import fused_gatconv as gat
    ar = torch.randn(m, h, generator=gen).to(DEV) * 4
    ac = torch.randn(m, h, generator=gen).to(DEV) * 4
    want_gat = oracle_mod.gat_forward(n_(row_ptr), n_(col_ind), n_(ar), n_(ac), 0.2, n_(V))
    assert _rel_to_max(gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, V), want_gat) <= 2e-5


def test_block_plan_mixed_fit_and_spill(oracle_mod):
    """A batch holding one graph too large for LDS: its rows take the general kernel (spill chunks), the
    other graphs the resident kernel; isolated nodes form their own closed ranges."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import Graph, batch
    from DFGNN.utils import synthetic as S
    rng = np.random.default_rng(3)
    big_n = 700
    iu, ju = np.triu_indices(big_n, k=1)
    keep = rng.random(len(iu)) < 0.05
    big = Graph(np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]]), big_n)
    lonely = Graph(np.zeros(0, np.int64), np.zeros(0, np.int64), 5)
    # 180 nodes at 55 % density: its K/V rows fit LDS but its per-edge array does not -> "edge-global" range
    iu, ju = np.triu_indices(180, k=1)
    keep = rng.random(len(iu)) < 0.55
    dense = Graph(np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]]), 180)
    g = batch([S.pattern_like(batch_size=3, seed=1), big, lonely, dense, S.pattern_like(batch_size=2, seed=2)]).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, 1, 128, seed=8, device=DEV)
    out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    plan = row_ptr._dfgnn_plans[128]
    assert plan.num_fit >= 3 and plan.num_spill == (big_n + 15) // 16 and plan.num_edge_global == 1
    inf = gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0]   # inference path needs the scratch
    assert torch.allclose(inf, out, atol=1e-6)
    n = lambda t: t.cpu().numpy()  # noqa: E731
    want, want_attn = oracle_mod.gt_forward(n(row_ptr), n(col_ind), n(val), n(Q), n(K), n(V), want_attn=True)
    _close(out, want, "mixed fit/spill out")
    _close(attn, want_attn, "mixed fit/spill attn")
    dO = torch.randn_like(out)
    wq, wk, wv = oracle_mod.gt_backward(n(row_ptr), n(col_ind), n(val), n(Q), n(K), n(V), n(dO))
    dQ, dK, dV = gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO)
    _close(dQ, wq, "mixed dQ")
    _close(dK, wk, "mixed dK")
    _close(dV, wv, "mixed dV")
    import fused_gatconv as gat
    ar, ac, X = S.gat_features(m, 1, 128, seed=6, device=DEV)
    want_gat = oracle_mod.gat_forward(n(row_ptr), n(col_ind), n(ar), n(ac), 0.2, n(X))
    _close(gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X), want_gat, "mixed GAT hyper")


# ---- GAT training pair: gat_forward / gat_backward behind FusedGATFunction -------------
def _gat_train_case(oracle_mod, g, attn_drop, seed=0):
    """Run GATConvFuse fwd+bwd on CSR/CSC arrays of dict g (numpy); the dropout randoms drawn by the binding are
    read back and handed to the oracle, so both sides drop the same edges."""
    import fused_gatconv as gat
    from DFGNN.operators.fused_gatconv import GATConvFuse
    ip, idx = _t(g["row_ptr"]), _t(g["col_ind"])
    cp, ri, pm = _t(g["col_ptr"]), _t(g["row_ind"]), _t(g["val_idx"])
    slope = float(g["negative_slope"])
    ar, ac, X = (_t(g[k]).requires_grad_(True) for k in ("attn_row", "attn_col", "V"))
    dO = _t(g["dO"])
    torch.manual_seed(seed)
    out, emax, esum, mask = gat.gat_forward(ar.detach(), ac.detach(), ip, idx, slope, X.detach(), attn_drop)
    assert tuple(mask.shape) == (len(g["col_ind"]), X.shape[1])
    mask_np = mask.cpu().numpy() if attn_drop > 0 else None
    args = (g["row_ptr"], g["col_ind"], g["attn_row"], g["attn_col"], slope, g["V"])
    w_out, w_max, w_sum = oracle_mod.gat_train_forward(*args, mask_np, attn_drop)
    w_gf, w_gr, w_gc = oracle_mod.gat_backward(*args, g["dO"], mask_np, attn_drop)
    errs = dict(out=_close(out, w_out, "gat_forward out"), sum=_close(esum, w_sum, "edge_sum"))
    deg = np.diff(g["row_ptr"])
    _close(emax[torch.from_numpy(deg > 0).to(DEV)], w_max[deg > 0], "edge_max")
    assert bool((emax[torch.from_numpy(deg == 0).to(DEV)] < -9e37).all())
    gf, gr, gc = gat.gat_backward(slope, attn_drop, ip, idx, cp, ri, pm, emax, esum, mask, X.detach(), ar.detach(),
                                  ac.detach(), dO)
    errs.update(grad_feat=_close(gf, w_gf, "grad_feat"), grad_row=_close(gr, w_gr, "grad_attn_row"),
                grad_col=_close(gc, w_gc, "grad_attn_col"))
    # and through the autograd.Function, same seed -> same randoms
    torch.manual_seed(seed)
    o2 = GATConvFuse(ar, ac, ip, idx, cp, ri, pm, slope, X, attn_drop)
    o2.backward(dO)
    assert torch.equal(o2.detach(), out)
    assert torch.equal(X.grad, gf) and torch.equal(ar.grad, gr) and torch.equal(ac.grad, gc)
    return errs


def test_gat_train_golden(oracle_mod, golden, golden_gat_train):
    """Committed fixtures (no dropout: the binding draws its own randoms, so the *_drop expectations are checked
    through the C ABI with the fixture's mask below)."""
    import dfgnn_native as n
    for name, t in golden_gat_train.items():
        g = golden[name]
        errs = _gat_train_case(oracle_mod, g, 0.0)
        ip, idx = _t(g["row_ptr"]), _t(g["col_ind"])
        cp, ri, pm = _t(g["col_ptr"]), _t(g["row_ind"]), _t(g["val_idx"])
        ar, ac, X, dO, mask = _t(g["attn_row"]), _t(g["attn_col"]), _t(g["V"]), _t(g["dO"]), _t(t["edge_mask"])
        m, h, f = X.shape
        nnz, drop, slope = idx.numel(), float(t["attn_drop"]), float(g["negative_slope"])
        out, gf = torch.empty_like(X), torch.empty_like(X)
        emax, esum, gr, gc = (torch.empty(m, h, device=DEV) for _ in range(4))
        ws = torch.empty(h, nnz, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        p = lambda x: x.data_ptr()  # noqa: E731
        n.check(n.lib().dfgnn_gat_fwd_train(m, nnz, h, f, p(ip), p(idx), None, p(ar), p(ac), slope, p(X), p(mask), drop,
                                            p(emax), p(esum), p(out), None, None, st), "fwd")
        n.check(n.lib().dfgnn_gat_bwd(m, nnz, h, f, p(ip), p(idx), None, p(cp), p(ri), p(pm), p(ar), p(ac), slope,
                                      p(X), p(emax), p(esum), p(mask), drop, p(dO), p(ws), p(gf), p(gr), p(gc), None,
                                      None, st), "bwd")
        for got, key in ((out, "out"), (esum, "edge_sum"), (gf, "grad_feat"), (gr, "grad_attn_row"),
                         (gc, "grad_attn_col")):
            errs[key + "_drop"] = _close(got, t[key + "_drop"], f"{name}:{key}_drop")
        print(name, {k: f"{v:.1e}" for k, v in errs.items()})
    _loaded_native()


@pytest.mark.parametrize("m,avg,h,f,kw", [
    (1, 0, 1, 4, {}),
    (5, 0, 2, 8, {}),
    (333, 9, 1, 128, dict(empty_frac=0.1)),
    (257, 40, 1, 64, dict(dup_frac=0.3)),
    (64, 4, 8, 16, dict(empty_frac=0.5)),
    (100, 30, 4, 32, {}),
    (40, 6, 2, 256, {}),
    (24, 5, 1, 512, {}),
    (50, 7, 2, 7, {}),
    (30, 6, 2, 130, {}),
    (300, 12, 1, 128, dict(max_deg=5000)),
    (40, 300, 1, 32, {}),
])
@pytest.mark.parametrize("attn_drop", [0.0, 0.3])
def test_gat_train_random_graphs(oracle_mod, m, avg, h, f, kw, attn_drop):
    g = _random_case(oracle_mod, 11 * m + f, m, avg, h, f, **kw)
    print(_gat_train_case(oracle_mod, g, attn_drop, seed=m))


def test_gat_train_pattern_like_batch(oracle_mod):
    """The C3 shape (PATTERN-like batch, f = 128) at 64 graphs, through preprocess_Hyper_fw_bw's CSC arrays."""
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = S.pattern_like(batch_size=64, seed=1).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=DEV)
    dO = torch.randn(m, 1, 128, generator=torch.Generator().manual_seed(2)).to(DEV)
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    case = dict(row_ptr=n_(row_ptr), col_ind=n_(col_ind), col_ptr=n_(col_ptr), row_ind=n_(row_ind),
                val_idx=n_(val_idx), attn_row=n_(ar), attn_col=n_(ac), V=n_(X), dO=n_(dO),
                negative_slope=np.float32(0.2))
    for drop in (0.0, 0.5):
        print(drop, _gat_train_case(oracle_mod, case, drop))


@pytest.mark.parametrize("attn_drop", [0.0, 0.4])
@pytest.mark.parametrize("h,f", [(1, 128), (2, 64), (3, 32), (4, 16), (2, 8)])
def test_gat_train_dense_every_geometry(oracle_mod, h, f, attn_drop):
    """GAT training pair on the matrix-core kernels (gat_dense_fwd_kernel with row statistics, gat_dense_bwd_kernel): a
    batch whose ranges are all dense and hit every geometry (1 strip, <= 128 nodes, 129-160 nodes with two row blocks,
    161-255 nodes with 2 x 2 tiles, isolated nodes, a directed graph whose last rows have in-edges only), against
    the oracle and against the general CSR kernels."""
    import fused_gatconv as gat
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import Graph, batch
    from DFGNN.utils import synthetic as S
    rng = np.random.default_rng(23 + f)

    def er(n, p, drop_rows=()):
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(len(iu)) < p
        keep &= ~np.isin(iu, drop_rows) & ~np.isin(ju, drop_rows)
        return np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]])

    graphs = []
    for n, p in ((9, 0.9), (17, 0.6), (64, 0.5), (128, 0.35), (129, 0.3), (145, 0.4), (160, 0.3), (161, 0.3), (200, 0.25),
                 (255, 0.2)):
        s_, d_ = er(n, p)
        graphs.append(Graph(s_, d_, n))
    s_, d_ = er(70, 0.5, drop_rows=(0, 33, 69))
    graphs.append(Graph(s_, d_, 70))
    keep = rng.random((80, 140)) < 0.3
    ds_, dd_ = np.nonzero(keep)
    graphs.append(Graph(ds_.astype(np.int64), dd_.astype(np.int64), 140))
    g = batch(graphs).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    ar, ac, X = S.gat_features(m, h, f, seed=8, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    torch.manual_seed(77)
    out, emax, esum, mask = gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, attn_drop)
    plan = row_ptr._dfgnn_plans[f]
    assert plan.num_dense == plan.num_fit > 0 and plan.num_spill == 0      # -> the library takes the dense kernels
    assert torch.equal(row_ptr._dfgnn_rows[1], rows)                        # derived COO rows == preprocess_Hyper's
    gf, gr, gc = gat.gat_backward(0.2, attn_drop, row_ptr, col_ind, col_ptr, row_ind, val_idx, emax, esum, mask, X, ar,
                                  ac, dO)
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    args = (n_(row_ptr), n_(col_ind), n_(ar), n_(ac), 0.2, n_(X))
    mask_np = n_(mask) if attn_drop > 0 else None
    w_out, w_max, w_sum = oracle_mod.gat_train_forward(*args, mask_np, attn_drop)
    w_gf, w_gr, w_gc = oracle_mod.gat_backward(*args, n_(dO), mask_np, attn_drop)
    deg = np.diff(n_(row_ptr))
    has = torch.from_numpy(deg > 0).to(DEV)
    errs = dict(out=_close(out, w_out, "dense gat_forward out"), sum=_close(esum, w_sum, "dense edge_sum"),
                max=_close(emax[has], w_max[deg > 0], "dense edge_max"), gf=_close(gf, w_gf, "dense grad_feat"),
                gr=_close(gr, w_gr, "dense grad_attn_row"), gc=_close(gc, w_gc, "dense grad_attn_col"))
    assert bool((emax[~has] < -9e37).all())
    print({k: f"{v:.1e}" for k, v in errs.items()})
    gat.USE_BLOCK_PLAN = False
    try:
        torch.manual_seed(77)                                                # the same dropout randoms
        out_n, emax_n, esum_n, mask_n = gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, attn_drop)
        assert torch.equal(mask_n, mask)
        gf_n, gr_n, gc_n = gat.gat_backward(0.2, attn_drop, row_ptr, col_ind, col_ptr, row_ind, val_idx, emax_n, esum_n,
                                            mask_n, X, ar, ac, dO)
    finally:
        gat.USE_BLOCK_PLAN = True
    for a, b in ((out, out_n), (gf, gf_n), (gr, gr_n), (gc, gc_n), (esum, esum_n)):
        assert torch.allclose(a, b, atol=2e-4, rtol=1e-3)
    assert not torch.equal(out, out_n)                                       # two different code paths did run
    # mixed batch: dense ranges on the matrix cores, the rest -- a graph with a duplicate edge (fit, not dense), a graph
    # too large for LDS residency (spill chunks), isolated nodes -- on the general kernels restricted to those ranges
    s_, d_ = er(40, 0.6)
    big_n = 1500
    bs_, bd_ = rng.integers(0, big_n, 40000), rng.integers(0, big_n, 40000)
    g2 = batch(graphs[:3] + [Graph(np.concatenate([s_, s_[:1]]), np.concatenate([d_, d_[:1]]), 40),
                             Graph(bs_, bd_, big_n)] + graphs[3:6]).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g2)
    m2 = g2.num_nodes()
    ar, ac, X = S.gat_features(m2, h, f, seed=9, device=DEV)
    dO = torch.randn(m2, h, f, generator=torch.Generator().manual_seed(4)).to(DEV)
    torch.manual_seed(5)
    out2, emax2, esum2, mask2 = gat.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, attn_drop)
    p2 = row_ptr._dfgnn_plans[f]
    assert 0 < p2.num_dense < p2.num_fit and (p2.num_spill > 0 or f < 32)   # (narrow rows: the big graph still fits LDS)
    gf2, gr2, gc2 = gat.gat_backward(0.2, attn_drop, row_ptr, col_ind, col_ptr, row_ind, val_idx, emax2, esum2, mask2, X,
                                     ar, ac, dO)
    args2 = (n_(row_ptr), n_(col_ind), n_(ar), n_(ac), 0.2, n_(X))
    mk2 = n_(mask2) if attn_drop > 0 else None
    w2, _, ws2 = oracle_mod.gat_train_forward(*args2, mk2, attn_drop)
    wf2, wr2, wc2 = oracle_mod.gat_backward(*args2, n_(dO), mk2, attn_drop)
    for got, want, what in ((out2, w2, "out"), (esum2, ws2, "edge_sum"), (gf2, wf2, "grad_feat"),
                            (gr2, wr2, "grad_attn_row"), (gc2, wc2, "grad_attn_col")):
        _close(got, want, "mixed batch " + what)


# ---- GPU-side preprocessing: COO -> CSR / rows / CSC through dfgnn_preprocess_hyper ----------
@pytest.mark.parametrize("case", ["pattern", "random_dups", "isolated_tail", "int32_ids", "one_heavy_row", "no_edges",
                                  "single_node_loop"])
def test_native_preprocess_matches_torch_path(oracle_mod, case):
    """Bit-exact (integer work): the native GPU preprocessing against the numpy oracle of the reference's
    preprocess_Hyper_fw_bw arrays (oracle.coo_to_hyper, itself checked against scipy in tests/test_host.py) and against
    the torch path the same functions take for CPU graphs."""
    import dfgnn_preprocess
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.layers.util import preprocess_CSR, preprocess_Hyper, preprocess_softmax
    from DFGNN.utils import Graph
    from DFGNN.utils import synthetic as S
    rng = np.random.default_rng(len(case))
    dtype = torch.int64
    if case == "pattern":
        g = S.pattern_like(batch_size=48, seed=3)
        src, dst = (t.numpy() for t in g.edges())
        n = g.num_nodes()
    elif case == "random_dups":
        n = 700
        src, dst = rng.integers(0, n, 20000), rng.integers(0, n, 20000)
        src[5000:6000], dst[5000:6000] = src[:1000], dst[:1000]          # duplicate edges, far apart in COO order
    elif case == "isolated_tail":
        n = 5000
        src, dst = rng.integers(0, 40, 3000), rng.integers(0, 40, 3000)  # nodes 40.. have no edges at all
    elif case == "int32_ids":
        n = 300
        src, dst = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
        dtype = torch.int32
    elif case == "one_heavy_row":
        n = 1000
        src = np.concatenate([np.full(50000, 7), rng.integers(0, n, 2000)])
        dst = np.concatenate([rng.integers(0, n, 50000), np.full(2000, 999)])
    elif case == "no_edges":
        n = 17
        src = dst = np.zeros(0, np.int64)
    else:
        n = 1
        src = dst = np.zeros(3, np.int64)
    want = oracle_mod.coo_to_hyper(src, dst, n)
    ts, td = torch.from_numpy(src).to(dtype).to(DEV), torch.from_numpy(dst).to(dtype).to(DEV)
    got = dfgnn_preprocess.coo_to_hyper(ts, td, n, csc=True)
    for t, key in zip(got, ("row_ptr", "col_ind", "rows", "edge_order", "col_ptr", "row_ind", "val_idx")):
        assert t.dtype == torch.int32
        np.testing.assert_array_equal(t.cpu().numpy(), want[key], err_msg=f"{case}:{key}")
    half = dfgnn_preprocess.coo_to_hyper(ts, td, n, csc=False)
    assert len(half) == 4 and all(torch.equal(a, b) for a, b in zip(half, got[:4]))
    # the preprocess_* functions: GPU graph (native) == CPU graph (torch restatement of dgl.sparse)
    gc = Graph(src, dst, n)
    gg = gc.to(DEV)
    for fn in (preprocess_CSR, preprocess_Hyper, preprocess_softmax, preprocess_Hyper_fw_bw):
        a, b = fn(gg), fn(gc)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            if torch.is_tensor(x):
                assert x.dtype == y.dtype and torch.equal(x.cpu(), y), (case, fn.__name__)
            elif isinstance(x, int):
                assert x == y
    _loaded_native()


@pytest.mark.parametrize("shape", ["peptides", "pattern"])
def test_hipgraph_capture_of_a_training_step(shape):
    """The C ABI is capturable (no allocation / host sync inside, launches on the caller's stream): a forward +
    backward pair of operator calls recorded into a HIP graph replays bit-identically, also after the static inputs
    were overwritten in place."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import GraphedStep
    from DFGNN.utils import synthetic as S
    if shape == "peptides":
        g, h, f = S.peptides_like(batch_size=32, seed=3).to(DEV), 4, 32      # low-degree kernels
    else:
        g, h, f = S.pattern_like(batch_size=24, seed=2).to(DEV), 1, 128      # plan + matrix-core kernels
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, h, f, seed=3, device=DEV)
    dO = torch.randn(m, h, f, device=DEV)

    def step():
        out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        return [out] + gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO)

    eager = [t.clone() for t in step()]
    graphed = GraphedStep(step)
    for a, b in zip(eager, graphed.replay()):
        assert torch.equal(a, b)
    Q.mul_(0.5)                                            # next "batch" of features, same structure
    V.add_(1.0)
    again = [t.clone() for t in graphed.replay()]
    for a, b in zip(step(), again):
        assert torch.equal(a, b)
    assert not torch.equal(again[0], eager[0])


def test_torch_extension_and_ctypes_bindings_agree():
    """The torch C++ extension (csrc/torch_ext.cpp) and the ctypes binding end in the same C ABI calls: bit-identical
    results for every entry point the extension serves, the same RuntimeError for a bad argument, and the extension is
    the one the operator modules use."""
    import dfgnn_native
    import fused_gatconv as gat
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    assert dfgnn_native.ext() is not None
    g = S.pattern_like(batch_size=12, seed=8).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, 2, 64, seed=1, device=DEV)
    dO = torch.randn_like(Q)
    ar, ac, X = S.gat_features(m, 2, 64, seed=2, device=DEV)

    def run():
        out, attn = gt.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        res = [out, attn] + list(gt.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, attn, dO))
        res += gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)
        res += [gat.gat_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X),
                gat.gat_inference_softmax(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X),
                gat.gat_inference_softmax_gm(ar, ac, row_ptr, col_ind, rows, 0.2, X),
                gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)]
        try:
            gt.gt_hyper_inference(row_ptr, col_ind, rows.long(), val, smem, Q, K, V)
            err = None
        except RuntimeError as e:
            err = str(e)
        return res, err

    via_ext, err_ext = run()
    saved = dfgnn_native._ext
    dfgnn_native._ext = None                      # force the ctypes path
    try:
        via_ctypes, err_ctypes = run()
    finally:
        dfgnn_native._ext = saved
    assert len(via_ext) == len(via_ctypes) == 10
    for a, b in zip(via_ext, via_ctypes):
        assert torch.equal(a, b)
    assert err_ext and err_ctypes and "int32" in err_ext and "int32" in err_ctypes


def test_hipgraph_refuses_autograd_callables():
    """GraphedStep captures explicit operator calls; a callable that runs the autograd engine (the round-1 crash in
    hipStreamEndCapture, DFGNN/utils/hipgraph.py) is refused with a RuntimeError before any capture starts."""
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
    from DFGNN.utils import GraphedStep
    from DFGNN.utils import synthetic as S
    g = S.pattern_like(batch_size=4, seed=2).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    Q, K, V = (t.requires_grad_(True) for t in S.gt_features(g.num_nodes(), 1, 64, seed=1, device=DEV))
    dO = torch.randn_like(Q)

    def autograd_step():
        out = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        return torch.autograd.grad(out, (Q, K, V), dO)

    with pytest.raises(RuntimeError, match="explicit operator calls"):
        GraphedStep(autograd_step)
    assert not torch.cuda.is_current_stream_capturing()
    out = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Q, K, V)   # the device is fine
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("h,f,batched", [(1, 128, True), (4, 32, True), (2, 20, False), (3, 7, False)])
def test_gat_hyper_v2_and_recompute(oracle_mod, h, f, batched):
    """gat_inference_hyper_v2 (scores kernel + conv, multi-head a_l / a_r handed over as the
    layers do: a transposed view) and gat_inference_hyper_recompute against the oracle."""
    import fused_gatconv as gat
    from DFGNN.layers.util import preprocess_Hyper
    from DFGNN.utils import Graph
    from DFGNN.utils import synthetic as S
    rng = np.random.default_rng(h * 100 + f)
    if batched:
        g = S.pattern_like(batch_size=12, seed=5).to(DEV)
    else:
        n = 300
        g = Graph(rng.integers(0, n, 4000), rng.integers(0, n, 4000), n).to(DEV)
    indptr, indices, rows, val, smem = preprocess_Hyper(g)
    m = g.num_nodes()
    X = torch.randn(m, h, f, device=DEV)
    a_l = torch.randn(1, f, h, device=DEV).transpose(1, 2)         # [1, h, f] view, not contiguous for h > 1
    a_r = torch.randn(1, f, h, device=DEV).transpose(1, 2)
    ar, ac = (a_l * X).sum(-1), (a_r * X).sum(-1)
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want = oracle_mod.gat_forward(n_(indptr), n_(indices), n_(ar), n_(ac), 0.2, n_(X))
    _close(gat.gat_inference_hyper_v2(smem, a_l, a_r, indptr, indices, 0.2, X), want, "hyper_v2")
    _close(gat.gat_inference_hyper_recompute(ar, ac, indptr, indices, 0.2, X), want, "hyper_recompute")
    # the scores kernel on its own, against float64
    L = __import__("dfgnn_native").lib()
    sr, sc = torch.empty(m, h, device=DEV), torch.empty(m, h, device=DEV)
    al, arr = a_l.reshape(h, f).contiguous(), a_r.reshape(h, f).contiguous()
    assert L.dfgnn_gat_attn_scores(m, h, f, al.data_ptr(), arr.data_ptr(), X.data_ptr(), sr.data_ptr(), sc.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream) == 0
    _close(sr, n_((al.double() * X.double()).sum(-1)), "attn_row scores")
    _close(sc, n_((arr.double() * X.double()).sum(-1)), "attn_col scores")


def test_agnn_training_layer_grads_match_autograd_baseline():
    """AGNNConv_forward: projection-weight gradients of the fused path (GTConvFuse_hyper with
    Q = K = normalised H, V = H) vs autograd through the non-fused torch branch."""
    from DFGNN.layers import AGNNConv_forward, preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    torch.manual_seed(2)
    g = S.pattern_like(batch_size=8, seed=6).to(DEV)
    params = preprocess_Hyper_fw_bw(g)
    layer = AGNNConv_forward(64, 64, 1).to(DEV).train()
    x = torch.randn(g.num_nodes(), 64, device=DEV)
    grads = []
    for fuse in (False, True):
        layer.zero_grad()
        out = layer(params, x, fuse=fuse)
        (out * torch.linspace(-1, 1, out.numel(), device=DEV).view_as(out)).sum().backward()
        grads.append((out.detach().clone(), layer.proj.weight.grad.clone(), layer.proj.bias.grad.clone()))
    for a, b, what in zip(grads[0], grads[1], ("out", "proj.weight.grad", "proj.bias.grad")):
        scale = float(a.abs().max())
        assert torch.allclose(a, b, atol=1e-3 * max(1.0, scale), rtol=1e-3), what


def test_multilayer_training_fused_vs_baseline():
    """tools/train_stack.py at toy size: a 3-layer residual stack trained with Adam on fresh batches, fused operators
    vs the non-fused torch branch -- the losses must track each other (same init, same data)."""
    import argparse
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("train_stack", f"{ROOT}/tools/train_stack.py")
    ts = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ts)
    args = argparse.Namespace(layers=3, batch_size=6, dim=64, steps=5, warmup=1, batches=2)
    fused = ts.run(args, True, torch.device(DEV))
    base = ts.run(args, False, torch.device(DEV))
    assert abs(fused["final_loss"] - base["final_loss"]) <= 1e-3 * max(1.0, abs(base["final_loss"]))
    # the trained q_proj weights: five Adam steps through the fused gradients vs through autograd of the torch branch
    # (Adam normalises the gradient, so compare the weights loosely and their movement from a common init tightly)
    # (an element whose gradient is ~0 can take Adam steps of opposite sign on the two paths: judge the bulk, and bound
    # the rest by what lr * steps allows)
    diff = (fused["qkv_weights"] - base["qkv_weights"]).abs()
    assert float(diff.mean()) < 2e-4 and float(torch.quantile(diff, 0.999)) < 2e-3 and float(diff.max()) < 2e-2
    assert fused["preprocess_ms"] > 0 and fused["layer_edges_per_s"] > 0


def _reference_plan(row_ptr, col_ind, f, budget, merge_nodes):
    """numpy restatement of dfgnn_plan_build's semantics (plan.hip): natural closed ranges, greedy left-to-right merge
    under the LDS budget / merge cap, classes (spill / edge-global / dense).  Returns (fit list in node order with
    flags, spill chunks)."""
    m = len(row_ptr) - 1
    ends = _natural_ranges(row_ptr, col_ind)
    dup = np.zeros(m + 1, np.int64)                     # rows with a duplicate edge (or too wide / long) before row i
    for i in range(m):
        c = col_ind[row_ptr[i]:row_ptr[i + 1]]
        lo, hi = (min(i, c.min()), max(i, c.max())) if len(c) else (i, i)
        bad = 1 if (hi - lo >= 256 or len(c) >= 256) else int(len(np.unique(c)) != len(c))
        dup[i + 1] = dup[i] + bad
    lite = lambda n, e: min(n, 1 << 16) * (4 * f + 8) + min(e, 1 << 24) * (1 if min(n, 1 << 16) <= 256 else 2)  # noqa: E731
    full = lambda n, e: lite(n, e) + 4 * min(e, 1 << 24)  # noqa: E731
    dense_f = f in (8, 16, 32, 64, 128)
    fit, spill = [], []

    def flush(n0, n1):
        if n1 <= n0:
            return
        e, nn = int(row_ptr[n1] - row_ptr[n0]), n1 - n0
        flags = ((1 << 30) if full(nn, e) > budget else 0)
        if dense_f and nn <= 255 and min(e, 1 << 24) * 32 >= nn * nn and dup[n1] == dup[n0]:
            flags |= 1 << 29
        fit.append((n0, n1 | flags))

    cur0 = cur1 = prev = 0
    for end in ends:
        n_one, e_one = end - prev, int(row_ptr[end] - row_ptr[prev])
        if lite(n_one, e_one) > budget:
            flush(cur0, cur1)
            spill += [(r, min(end, r + 16)) for r in range(prev, end, 16)]
            cur0 = cur1 = end
        elif full(n_one, e_one) > budget:
            flush(cur0, cur1)
            flush(prev, end)
            cur0 = cur1 = end
        elif cur1 > cur0 and (full(end - cur0, int(row_ptr[end] - row_ptr[cur0])) > budget or end - cur0 > merge_nodes):
            flush(cur0, cur1)
            cur0, cur1 = prev, end
        else:
            if cur1 == cur0:
                cur0 = prev
            cur1 = end
        prev = end
    flush(cur0, cur1)
    return fit, spill


@pytest.mark.parametrize("case", ["tiny_graphs_merge", "many_ranges_serial_fallback", "mixed_classes", "duplicates"])
def test_block_plan_matches_reference_walk(case):
    """The parallel merge of plan_cut_kernel (and its serial fallback beyond 4096 natural ranges) against a numpy
    restatement of the greedy walk: same fit ranges with the same flags, same spill chunks."""
    from _binding_util import build_plan
    from DFGNN.layers.util import preprocess_Hyper
    from DFGNN.utils import Graph, batch
    rng = np.random.default_rng(len(case))

    def clique(n):
        i, j = np.nonzero(~np.eye(n, dtype=bool))
        return Graph(i, j, n)

    def er(n, p):
        iu, ju = np.triu_indices(n, k=1)
        keep = rng.random(len(iu)) < p
        return Graph(np.concatenate([iu[keep], ju[keep]]), np.concatenate([ju[keep], iu[keep]]), n)

    f = 128
    if case == "tiny_graphs_merge":                      # 600 graphs of 9-20 nodes: ~8 merge into each <= 128-node range
        graphs = [clique(int(n)) for n in rng.integers(9, 21, 600)]
    elif case == "many_ranges_serial_fallback":           # 5000 natural ranges > 4096
        graphs = [clique(int(n)) for n in rng.integers(9, 12, 5000)]
    elif case == "mixed_classes":                         # a range that does not fit (spill), one whose edge array does
        graphs = [er(40, 0.5), er(700, 0.3), er(60, 0.6), er(250, 0.95), clique(12), clique(30), er(180, 0.5)]
        f = 64                                            # not (edge-global), merged small ones, dense ones
    else:
        g1, g2 = er(50, 0.5), er(70, 0.5)
        s_, d_ = g1.edges()
        graphs = [Graph(torch.cat([s_, s_[:3]]), torch.cat([d_, d_[:3]]), 50), g2, clique(10), clique(11)]
    g = batch(graphs).to(DEV)
    row_ptr, col_ind, rows, val, smem = preprocess_Hyper(g)
    plan = build_plan(row_ptr, col_ind, f)
    rp, ci = row_ptr.cpu().numpy(), col_ind.cpu().numpy()
    budget = plan.meta[7]
    want_fit, want_spill = _reference_plan(rp, ci, f, budget, 128)
    m = g.num_nodes()
    buf = plan.buf.cpu().numpy()
    nfit, nspill = plan.meta[0], plan.meta[1]
    got_fit = sorted(map(tuple, buf[12:12 + 2 * nfit].reshape(-1, 2).tolist()))
    got_spill = sorted(map(tuple, buf[12 + 2 * m:12 + 2 * m + 2 * nspill].reshape(-1, 2).tolist()))
    assert got_fit == sorted(want_fit), (case, len(got_fit), len(want_fit))
    assert got_spill == sorted(want_spill), case
    FLAGS = (1 << 30) | (1 << 29)
    n_dense = sum(1 for _, b in want_fit if b & (1 << 29))
    assert plan.meta[8] == sum(1 for _, b in want_fit if b & (1 << 30))
    assert plan.num_dense == (n_dense if nfit <= 4096 else 0)
    assert plan.meta[2] == max((b & ~FLAGS) - a for a, b in want_fit)
    if case == "mixed_classes":
        assert nspill > 0 and plan.meta[8] > 0 and n_dense > 0
    if case == "many_ranges_serial_fallback":
        assert len(_natural_ranges(rp, ci)) > 4096


def test_harness_scripts_run_like_the_reference(capsys):
    """DFGNN/script/test/test_batch_graph.py / test_full_graph.py with the reference's arguments, on the synthetic
    dataset stand-ins at toy size: every served format of a --format all sweep runs, prints the reference's verdict
    and result lines, and the fused path is the faster one on the batched GT case."""
    import argparse
    import os
    from DFGNN.script.harness import run_batch_graph, run_full_graph
    os.environ["DFGNN_SYNTH_GRAPHS"] = "24"
    ns = dict(config=None, heads=1, data_dir="./data", store_result=False, subgraph_filter=False, profile=False)
    res = run_batch_graph(argparse.Namespace(conv="gt", format="all", dim=64, batch_size=8, dataset="PATTERN", **ns))
    assert set(res) == {"csr", "softmax", "hyper"} and all(a > 0 and b > 0 for a, b in res.values())
    res = run_batch_graph(argparse.Namespace(conv="gat", format="hyper_v2", dim=64, batch_size=8, dataset="PATTERN", **ns))
    assert "hyper_v2" in res
    res = run_full_graph(argparse.Namespace(conv="gat", format="all", dim=64, batch_size=None, dataset="cora", **ns))
    assert set(res) == {"csr", "softmax", "hyper_v2", "tiling", "hyper_recompute"}
    res = run_full_graph(argparse.Namespace(conv="agnn", format="hyper", dim=64, batch_size=None, dataset="cora", **ns))
    out = capsys.readouterr().out
    assert "the results are the same, success!!!!!!!!!!" in out and "fuse average time" in out
    assert "mismatch" not in out    # the reference's own verdict (DFGNN/utils/util.py:211-236), every format


@pytest.mark.parametrize("h,f", [(8, 16), (4, 32), (2, 64), (2, 32), (12, 16), (4, 16)])
def test_multihead_forward_takes_heads_in_groups(oracle_mod, h, f):
    """Multi-head GT forward on the matrix cores (dfgnn_dense_heads.hpp: one workgroup per range, heads in groups of 64
    feature columns): every range class in one batch -- <= 128 nodes (one strip per wave), 129..160 (two strips, images
    fetched at the group boundary), > 160 (the per-head body), a 16-node graph, a strip with more edges than the per-wave
    staging area holds (160 nodes at 95 % density) -- training (attn_edge written) and inference, against the oracle;
    features of very different magnitude per head, so a group's shared image scale is exercised."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = _dense_batch([(16, 0.6), (100, 0.45), (128, 0.5), (129, 0.4), (150, 0.45), (160, 0.95), (161, 0.3), (200, 0.3),
                      (97, 0.1), (128, 1.0)], seed=4)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, h, f, seed=5, device=DEV)
    mag = torch.logspace(-2, 2, h, device=DEV).view(1, h, 1)          # head k of V is 10^(4 k / (h - 1) - 2) times head 0's
    V = (V * mag).contiguous()
    K = (K * torch.logspace(0.5, -0.5, h, device=DEV).view(1, h, 1)).contiguous()
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    plan = row_ptr._dfgnn_plans[f]
    assert plan.num_dense == plan.num_fit >= 9 and plan.num_spill == 0    # (the 16-node graph is merged with its neighbour)
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want, want_attn = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), want_attn=True)
    _close(attn, want_attn, f"h={h} f={f} attn_edge")
    inf = gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0]
    for k in range(h):  # per head, in units of the head's own magnitude: the heads differ by four orders of magnitude
        mk = float(mag[0, k, 0])
        _close(out[:, k] / mk, want[:, k] / mk, f"h={h} f={f} out, head {k}")
        _close(inf[:, k] / mk, want[:, k] / mk, f"h={h} f={f} inference, head {k}")
    # backward: ranges of <= 128 nodes take one workgroup that walks the heads (two image buffers, the tile cleared once:
    # dS is zero wherever P is), the larger ones one workgroup per head -- against the oracle and against the
    # per-(range, head) form of the same kernels
    Q, K, V = S.gt_features(m, h, f, seed=6, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(8)).to(DEV)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    dQ, dK, dV = gt.gt_backward(*args, attn, dO)
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    for got, ref, what in ((dQ, wq, "dQ"), (dK, wk, "dK"), (dV, wv, "dV")):
        _close(got, ref, f"h={h} f={f} {what}")


def test_bench_line_keeps_the_driver_contract():
    """`python bench.py --steps K --warmup W` prints ONE JSON line with the keys the driver and the judge read: the metric,
    whole-job value, the step time, `roofline` (bound / achieved / peak / unit / frac / traffic) and `cpu_baseline` (value /
    unit / cores / kind / sample), `config.workload`, no model keys.  Run as a child process, as the driver runs it."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--no-c4",
                          "--cpu-sample-graphs", "8"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "edges/s" and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.2 < r["frac"] < 1.0
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 1e8)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "edges/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] == pytest.approx(d["config"]["total_edges"] / (d["ms_per_step"] * 1e-3), rel=1e-3)


@pytest.mark.parametrize("heads", [1, 8])
def test_dense_kernels_write_nothing_outside_their_outputs(heads):
    """A poor man's sanitizer for the hand-written kernels (the pool has no GPU AddressSanitizer): every output of the
    forward and the backward -- out, attn_edge, dQ, dK, dV -- lives in the middle of a larger buffer filled with a bit
    pattern; after the launches (through the C ABI, as the bindings call it) the guard regions must be untouched and the
    outputs fully written (no element still carries the pattern).  Mixed batch: <= 128, 129..160 and > 160 nodes, one
    graph with more edges than the prefetch covers."""
    import dfgnn_native
    from _binding_util import get_plan
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    L = dfgnn_native.lib()
    g = _dense_batch([(100, 0.45), (128, 1.0), (129, 0.4), (150, 0.45), (160, 0.9), (161, 0.3), (200, 0.3), (33, 0.5)], seed=9)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    f = 128 // heads
    Q, K, V = S.gt_features(m, heads, f, seed=3, device=DEV)
    dO = torch.randn(m, heads, f, generator=torch.Generator().manual_seed(1)).to(DEV)
    plan, meta, _ = get_plan(row_ptr, col_ind, f, True)
    assert plan is not None and row_ptr._dfgnn_plans[f].num_dense == row_ptr._dfgnn_plans[f].num_fit
    GUARD, PAT = 1 << 18, -1.2345678e30            # floats on either side; a value no kernel produces

    def guarded(numel):
        buf = torch.full((numel + 2 * GUARD,), PAT, device=DEV)
        return buf, buf[GUARD:GUARD + numel]

    bufs = {k: guarded(n) for k, n in (("out", m * heads * f), ("attn", heads * nnz), ("dQ", m * heads * f),
                                        ("dK", m * heads * f), ("dV", m * heads * f), ("gedge", heads * nnz))}
    P = lambda t: t.data_ptr()  # noqa: E731
    stream = torch.cuda.current_stream().cuda_stream
    assert L.dfgnn_gt_hyper_fwd(m, nnz, heads, f, P(row_ptr), P(col_ind), P(rows), None, P(Q), P(K), P(V),
                                P(bufs["attn"][1]), None, P(bufs["out"][1]), plan, meta, stream) == 0
    assert L.dfgnn_gt_bwd(m, nnz, heads, f, P(row_ptr), P(col_ind), P(rows), None, P(col_ptr), P(row_ind), P(val_idx), P(Q),
                          P(K), P(V), P(bufs["attn"][1]), P(dO), P(bufs["gedge"][1]), P(bufs["dQ"][1]), P(bufs["dK"][1]),
                          P(bufs["dV"][1]), plan, meta, stream) == 0
    torch.cuda.synchronize()
    for name, (buf, view) in bufs.items():
        assert bool((buf[:GUARD] == PAT).all()) and bool((buf[-GUARD:] == PAT).all()), f"{name}: guard region written"
        if name != "gedge":                        # (the dense path keeps dS on chip: grad_edge is left untouched)
            assert not bool((view == PAT).any()), f"{name}: elements left unwritten"
            assert bool(torch.isfinite(view).all()), name


def test_gat_dense_kernels_write_nothing_outside_their_outputs():
    """The same guard-region check for the GAT training pair on a dense batch (gat_dense_fwd / bwd kernels, 2 heads)."""
    import dfgnn_native
    import fused_gatconv
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    L = dfgnn_native.lib()
    g = _dense_batch([(100, 0.45), (128, 1.0), (129, 0.4), (160, 0.9), (161, 0.3), (200, 0.3), (33, 0.5)], seed=10)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz, h, f = g.num_nodes(), g.num_edges(), 2, 64
    ar, ac, X = S.gat_features(m, h, f, seed=6, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(2)).to(DEV)
    g_rows, plan, meta = fused_gatconv._train_plan(row_ptr, col_ind, f, 0.0)
    assert plan is not None
    GUARD, PAT = 1 << 18, -1.2345678e30

    def guarded(numel):
        buf = torch.full((numel + 2 * GUARD,), PAT, device=DEV)
        return buf, buf[GUARD:GUARD + numel]

    bufs = {k: guarded(n) for k, n in (("out", m * h * f), ("emax", m * h), ("esum", m * h), ("gfeat", m * h * f),
                                        ("g_ar", m * h), ("g_ac", m * h), ("gedge", h * nnz))}
    P = lambda t: t.data_ptr()  # noqa: E731
    stream = torch.cuda.current_stream().cuda_stream
    assert L.dfgnn_gat_fwd_train(m, nnz, h, f, P(row_ptr), P(col_ind), P(g_rows), P(ar), P(ac), 0.2, P(X), None, 0.0,
                                 P(bufs["emax"][1]), P(bufs["esum"][1]), P(bufs["out"][1]), plan, meta, stream) == 0
    assert L.dfgnn_gat_bwd(m, nnz, h, f, P(row_ptr), P(col_ind), P(g_rows), P(col_ptr), P(row_ind), P(val_idx), P(ar), P(ac),
                           0.2, P(X), P(bufs["emax"][1]), P(bufs["esum"][1]), None, 0.0, P(dO), P(bufs["gedge"][1]),
                           P(bufs["gfeat"][1]), P(bufs["g_ar"][1]), P(bufs["g_ac"][1]), plan, meta, stream) == 0
    torch.cuda.synchronize()
    for name, (buf, view) in bufs.items():
        assert bool((buf[:GUARD] == PAT).all()) and bool((buf[-GUARD:] == PAT).all()), f"{name}: guard region written"
        if name != "gedge":
            assert not bool((view == PAT).any()), f"{name}: elements left unwritten"
            assert bool(torch.isfinite(view).all()), name


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_multihead_random_batches(oracle_mod, seed):
    """Randomised multi-head batches through the walking workgroups (forward groups of heads, backward head walk): graph
    sizes 16..200 and densities 0.08..1 drawn per graph, (heads, width) drawn per batch, training forward + backward and
    inference against the oracle.  The bodies hand LDS buffers from phase to phase on hand-placed barriers; odd shapes
    (strips past the range, a single strip, full density, staging areas that do not fit) are where a missing one shows."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    rng = np.random.default_rng(100 + seed)
    h, f = [(8, 16), (4, 32), (2, 64), (4, 16), (12, 16), (2, 32)][seed]
    sizes = [(int(rng.integers(16, 201)), float(rng.uniform(0.08, 1.0))) for _ in range(24)]
    sizes = [(n, max(p, 2.0 / n + 1.0 / 30)) for n, p in sizes]        # (dense enough for the matrix-core plan class)
    g = _dense_batch(sizes, seed=50 + seed)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, h, f, seed=7 + seed, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(seed)).to(DEV)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    plan = row_ptr._dfgnn_plans[f]
    assert plan.num_dense > 0
    dQ, dK, dV = gt.gt_backward(*args, attn, dO)
    inf = gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0]
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want, want_attn = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), want_attn=True)
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    for got, ref, what in ((out, want, "out"), (inf, want, "inference"), (attn, want_attn, "attn_edge"), (dQ, wq, "dQ"),
                           (dK, wk, "dK"), (dV, wv, "dV")):
        _close(got, ref, f"seed {seed} h={h} f={f} {what}")
    # twice more: the same launches must give the same bits (no dependence on workgroup timing)
    for _ in range(2):
        out2, attn2 = gt.gt_hyper_forward(*args)
        g2 = gt.gt_backward(*args, attn2, dO)
        assert torch.equal(out, out2) and torch.equal(attn, attn2)
        assert all(torch.equal(a, b) for a, b in zip((dQ, dK, dV), g2))


@pytest.mark.parametrize("h,f", [(8, 16), (8, 8), (4, 16)])
def test_multihead_small_width_on_matrix_cores(oracle_mod, h, f):
    """Multi-head GT with narrow heads (dim 128 / 8 heads, dim 64 / 8 heads): f = 16 and f = 8 run zero-padded on the
    32-wide matrix-core layout.  fwd + bwd against the oracle on an all-dense batch, and against the general kernels."""
    import fused_gtconv as gt
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.utils import synthetic as S
    g = S.pattern_like(batch_size=20, seed=11).to(DEV)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m = g.num_nodes()
    Q, K, V = S.gt_features(m, h, f, seed=5, device=DEV)
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(3)).to(DEV)
    args = (row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
    out, attn = gt.gt_hyper_forward(*args)
    plan = row_ptr._dfgnn_plans[f]
    assert plan.num_dense == plan.num_fit > 0 and plan.num_spill == 0
    dQ, dK, dV = gt.gt_backward(*args, attn, dO)
    n_ = lambda t: t.cpu().numpy()  # noqa: E731
    want, want_attn = oracle_mod.gt_forward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), want_attn=True)
    wq, wk, wv = oracle_mod.gt_backward(n_(row_ptr), n_(col_ind), n_(val), n_(Q), n_(K), n_(V), n_(dO))
    for got, ref, what in ((out, want, "out"), (attn, want_attn, "attn_edge"), (dQ, wq, "dQ"), (dK, wk, "dK"), (dV, wv, "dV")):
        _close(got, ref, f"h={h} f={f} {what}")
    _close(gt.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q, K, V)[0], want, "inference")
    gt.USE_BLOCK_PLAN = False
    try:
        out_n, attn_n = gt.gt_hyper_forward(*args)
    finally:
        gt.USE_BLOCK_PLAN = True
    assert torch.allclose(out, out_n, atol=2e-4, rtol=1e-3) and not torch.equal(out, out_n)   # two code paths
