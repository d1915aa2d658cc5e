"""CPU: the oracle against (a) the committed golden vectors, (b) the independent torch/autograd
restatement on fresh random graphs with the edge cases the kernels must survive, (c) hand-derived
known-answer cases.  The reference ships no vectors for this path (PARITY UNPINNED, oracle.c)."""
import numpy as np
import pytest
import torch

from conftest import random_graph
from oracle import torch_ref


def test_oracle_matches_golden(oracle_mod, golden):
    for name, g in golden.items():
        out, attn = oracle_mod.gt_forward(g["row_ptr"], g["col_ind"], g["val"], g["Q"], g["K"], g["V"], want_attn=True)
        dQ, dK, dV = oracle_mod.gt_backward(g["row_ptr"], g["col_ind"], g["val"], g["Q"], g["K"], g["V"], g["dO"])
        gat = oracle_mod.gat_forward(g["row_ptr"], g["col_ind"], g["attn_row"], g["attn_col"],
                                     float(g["negative_slope"]), g["V"])
        for got, key in ((out, "out"), (attn, "attn"), (dQ, "dQ"), (dK, "dK"), (dV, "dV"), (gat, "gat_out")):
            np.testing.assert_allclose(got, g[key], rtol=1e-6, atol=2e-6, err_msg=f"{name}:{key}")  # f32 storage


@pytest.mark.parametrize("m,avg,h,f,kw", [
    (1, 0, 1, 4, {}),                                   # single empty row
    (40, 3, 1, 16, dict(empty_frac=0.3)),               # isolated nodes
    (64, 12, 2, 32, dict(dup_frac=0.2)),                # duplicate edges
    (50, 5, 1, 128, dict(max_deg=300)),                 # degree > 64, > 128
    (30, 4, 3, 7, {}),                                  # odd feature width
])
def test_c_oracle_vs_torch_restatement(oracle_mod, m, avg, h, f, kw):
    rng = np.random.default_rng(m * 1000 + f)
    indptr, indices, _ = random_graph(rng, m, avg, **kw)
    nnz = len(indices)
    val = rng.uniform(0.5, 1.5, nnz).astype(np.float32)
    Q, K, V, dO = (rng.standard_normal((m, h, f)).astype(np.float32) for _ in range(4))
    out, attn = oracle_mod.gt_forward(indptr, indices, val, Q, K, V, want_attn=True)
    dQ, dK, dV = oracle_mod.gt_backward(indptr, indices, val, Q, K, V, dO)
    o2, a2, q2, k2, v2 = torch_ref.gt_forward_backward(indptr, indices, val, Q, K, V, dO)
    for a, b in ((out, o2), (attn, a2), (dQ, q2), (dK, k2), (dV, v2)):
        np.testing.assert_allclose(a, b.numpy(), rtol=0, atol=1e-9)
    ar, ac = (rng.standard_normal((m, h)).astype(np.float32) for _ in range(2))
    gat = oracle_mod.gat_forward(indptr, indices, ar, ac, 0.2, V)
    g2, _ = torch_ref.gat_forward(indptr, indices, ar, ac, 0.2, V)
    np.testing.assert_allclose(gat, g2.numpy(), rtol=0, atol=1e-9)
    # f32-accumulating build (the timed CPU baseline) stays within the parity tolerance of the f64 one
    out32 = oracle_mod.gt_forward(indptr, indices, val, Q, K, V, acc="f32")
    np.testing.assert_allclose(out32, out, rtol=0, atol=1e-3)


@pytest.mark.parametrize("m,avg,h,f,kw", [(40, 6, 2, 16, dict(empty_frac=0.2)), (64, 10, 1, 32, dict(dup_frac=0.25)),
                                         (33, 4, 3, 7, {})])
def test_c_oracle_vs_dense_masked_restatement(oracle_mod, m, avg, h, f, kw):
    """Third restatement (oracle/dense_ref.py: masked dense attention, closed-form softmax Jacobian, float64 numpy) --
    incl. duplicate edges (separate softmax terms) and empty rows."""
    from oracle import dense_ref
    rng = np.random.default_rng(7 * m + f)
    indptr, indices, _ = random_graph(rng, m, avg, **kw)
    val = np.ones(len(indices), np.float32)
    Q, K, V, dO = (rng.standard_normal((m, h, f)).astype(np.float32) for _ in range(4))
    out = oracle_mod.gt_forward(indptr, indices, val, Q, K, V)
    dQ, dK, dV = oracle_mod.gt_backward(indptr, indices, val, Q, K, V, dO)
    o3, q3, k3, v3 = dense_ref.gt_forward_backward(indptr, indices, Q, K, V, dO)
    for a, b in ((out, o3), (dQ, q3), (dK, k3), (dV, v3)):                # (the C oracle hands back fp32 arrays)
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)
    ar, ac = (rng.standard_normal((m, h)).astype(np.float32) for _ in range(2))
    np.testing.assert_allclose(oracle_mod.gat_forward(indptr, indices, ar, ac, 0.2, V),
                               dense_ref.gat_forward(indptr, indices, ar, ac, 0.2, V), rtol=1e-6, atol=1e-7)


def test_known_answers(oracle_mod):
    # (1) K == 0 -> uniform attention -> out = mean of neighbour V rows; empty row -> zeros
    indptr = np.array([0, 3, 3, 4], np.int32)
    indices = np.array([0, 1, 2, 1], np.int32)
    V = np.arange(3 * 1 * 2, dtype=np.float32).reshape(3, 1, 2)
    Q = np.ones_like(V)
    K = np.zeros_like(V)
    out, attn = oracle_mod.gt_forward(indptr, indices, np.ones(4, np.float32), Q, K, V, want_attn=True)
    np.testing.assert_allclose(out[0, 0], V[:, 0].mean(0))
    np.testing.assert_allclose(out[1, 0], 0.0)                 # fused_gtconv_hyper.cu:143
    np.testing.assert_allclose(out[2, 0], V[1, 0])             # single neighbour -> copy
    np.testing.assert_allclose(attn[0], [1 / 3, 1 / 3, 1 / 3, 1.0])
    # (2) two neighbours with logits ln(1), ln(3) -> P = (1/4, 3/4); val multiplies the logit
    indptr = np.array([0, 2, 2], np.int32)
    indices = np.array([0, 1], np.int32)
    Q = np.array([[[1.0]], [[0.0]]], np.float32)
    K = np.array([[[0.0]], [[np.log(3.0) / 2]]], np.float32)
    val = np.array([5.0, 2.0], np.float32)
    V = np.array([[[4.0]], [[8.0]]], np.float32)
    out = oracle_mod.gt_forward(indptr, indices, val, Q, K, V)
    np.testing.assert_allclose(out[0, 0, 0], 0.25 * 4 + 0.75 * 8, rtol=1e-6)
    # (3) GAT: LeakyReLU(0.2) of (-1, +1) -> logits (-0.2, 1)
    ar = np.array([[0.0], [0.0]], np.float32)
    ac = np.array([[-1.0], [1.0]], np.float32)
    gat = oracle_mod.gat_forward(indptr, indices, ar, ac, 0.2, V)
    p = np.exp([-0.2, 1.0]) / np.exp([-0.2, 1.0]).sum()
    np.testing.assert_allclose(gat[0, 0, 0], p[0] * 4 + p[1] * 8, rtol=1e-6)


def test_gat_train_oracle_matches_golden(oracle_mod, golden, golden_gat_train):
    for name, t in golden_gat_train.items():
        g = golden[name]
        args = (g["row_ptr"], g["col_ind"], g["attn_row"], g["attn_col"], float(g["negative_slope"]), g["V"])
        for sfx, mask, drop in (("", None, 0.0), ("_drop", t["edge_mask"], float(t["attn_drop"]))):
            out, emax, esum = oracle_mod.gat_train_forward(*args, mask, drop)
            gf, gr, gc = oracle_mod.gat_backward(*args, g["dO"], mask, drop)
            for got, key in ((out, "out"), (emax, "edge_max"), (esum, "edge_sum"), (gf, "grad_feat"),
                             (gr, "grad_attn_row"), (gc, "grad_attn_col")):
                np.testing.assert_allclose(got, t[key + sfx], rtol=2e-6, atol=4e-6, err_msg=f"{name}:{key}{sfx}")


@pytest.mark.parametrize("m,avg,h,f,kw", [
    (1, 0, 1, 4, {}),
    (40, 3, 2, 16, dict(empty_frac=0.3)),
    (64, 12, 2, 32, dict(dup_frac=0.2)),
    (50, 5, 1, 128, dict(max_deg=300)),
    (30, 4, 3, 7, {}),
])
def test_gat_train_oracle_vs_autograd(oracle_mod, m, avg, h, f, kw):
    """GAT training pair: hand-derived gradients of oracle.c vs autograd of the edge-list
    restatement, with and without attention dropout (same uniform randoms on both sides)."""
    rng = np.random.default_rng(m * 77 + f)
    indptr, indices, _ = random_graph(rng, m, avg, **kw)
    ar, ac = (rng.standard_normal((m, h)).astype(np.float32) for _ in range(2))
    X, dO = (rng.standard_normal((m, h, f)).astype(np.float32) for _ in range(2))
    for mask, drop in ((None, 0.0), (rng.random((len(indices), h)).astype(np.float32), 0.4)):
        out, emax, esum = oracle_mod.gat_train_forward(indptr, indices, ar, ac, 0.2, X, mask, drop)
        gf, gr, gc = oracle_mod.gat_backward(indptr, indices, ar, ac, 0.2, X, dO, mask, drop)
        o2, gf2, gr2, gc2 = torch_ref.gat_train(indptr, indices, ar, ac, 0.2, X, dO, mask, drop)
        for a, b in ((out, o2), (gf, gf2), (gr, gr2), (gc, gc2)):
            np.testing.assert_allclose(a, b.numpy(), rtol=0, atol=1e-9)
        if mask is None:
            np.testing.assert_allclose(out, oracle_mod.gat_forward(indptr, indices, ar, ac, 0.2, X), atol=1e-12)
        deg = np.diff(indptr)
        assert np.all(emax[deg == 0] < -9e37)                       # the reference's -1e38 sentinel
        assert np.all(esum[deg == 0] == 0) and np.all(esum[deg > 0] >= 1.0)
        # softmax gradients sum to zero over a row before the LeakyReLU factor: with slope == 1 grad_attn_row == 0
        _, gr1, gc1 = oracle_mod.gat_backward(indptr, indices, ar, ac, 1.0, X, dO, mask, drop)
        np.testing.assert_allclose(gr1, 0.0, atol=1e-10)


def test_gat_train_known_answer(oracle_mod):
    """One row, two neighbours, slope 0.2: pre = (-1, +1) -> s = (-0.2, 1); P = softmax(s);
    dP = dO * X_j; G = P (dP - <P, dP>) * (0.2, 1)."""
    indptr = np.array([0, 2, 2], np.int32)
    indices = np.array([0, 1], np.int32)
    ar = np.array([[0.0], [0.0]], np.float32)
    ac = np.array([[-1.0], [1.0]], np.float32)
    X = np.array([[[4.0]], [[8.0]]], np.float32)
    dO = np.array([[[0.5]], [[7.0]]], np.float32)
    p = np.exp([-0.2, 1.0]) / np.exp([-0.2, 1.0]).sum()
    out, emax, esum = oracle_mod.gat_train_forward(indptr, indices, ar, ac, 0.2, X)
    np.testing.assert_allclose(out[:, 0, 0], [p[0] * 4 + p[1] * 8, 0.0], rtol=1e-7)
    np.testing.assert_allclose(emax[0, 0], 1.0)
    np.testing.assert_allclose(esum[0, 0], 1.0 + np.exp(-1.2), rtol=1e-7)
    gf, gr, gc = oracle_mod.gat_backward(indptr, indices, ar, ac, 0.2, X, dO)
    dP = 0.5 * np.array([4.0, 8.0])
    G = p * (dP - (p * dP).sum()) * np.array([0.2, 1.0])
    np.testing.assert_allclose(gc[:, 0], G, rtol=1e-7)
    np.testing.assert_allclose(gr[:, 0], [G.sum(), 0.0], rtol=1e-7, atol=1e-15)
    np.testing.assert_allclose(gf[:, 0, 0], p * 0.5, rtol=1e-7)
    # dropout: edge 0 dropped at attn_drop 0.5 -> out = 2 p1 X1; grad_feat[0] = 0; g = (0, 2 dP1)
    mask = np.array([[0.1], [0.9]], np.float32)
    out, _, _ = oracle_mod.gat_train_forward(indptr, indices, ar, ac, 0.2, X, mask, 0.5)
    np.testing.assert_allclose(out[0, 0, 0], 2 * p[1] * 8, rtol=1e-7)
    gf, gr, gc = oracle_mod.gat_backward(indptr, indices, ar, ac, 0.2, X, dO, mask, 0.5)
    g = np.array([0.0, 2 * dP[1]])
    G = p * (g - (p * g).sum()) * np.array([0.2, 1.0])
    np.testing.assert_allclose(gc[:, 0], G, rtol=1e-7)
    np.testing.assert_allclose(gf[:, 0, 0], [0.0, 2 * p[1] * 0.5], rtol=1e-7)


def test_softmax_invariants(oracle_mod):
    """Properties the domain offers: rows of P sum to 1 (or 0 when empty); permuting a row's edges
    permutes P and leaves out unchanged; sum_i dV equals column sums of P weighted dO."""
    rng = np.random.default_rng(5)
    m, h, f = 37, 2, 8
    indptr, indices, rows = random_graph(rng, m, 6, empty_frac=0.2)
    nnz = len(indices)
    val = np.ones(nnz, np.float32)
    Q, K, V = (rng.standard_normal((m, h, f)).astype(np.float32) for _ in range(3))
    out, attn = oracle_mod.gt_forward(indptr, indices, val, Q, K, V, want_attn=True)
    deg = np.diff(indptr)
    rowsum = np.zeros((h, m))
    for hh in range(h):
        np.add.at(rowsum[hh], rows, attn[hh])
    np.testing.assert_allclose(rowsum, np.broadcast_to((deg > 0).astype(float), (h, m)), atol=1e-12)
    perm = np.arange(nnz)
    for i in range(m):
        seg = perm[indptr[i]:indptr[i + 1]]
        rng.shuffle(seg)
    out_p = oracle_mod.gt_forward(indptr, indices[perm], val, Q, K, V)
    np.testing.assert_allclose(out_p, out, atol=1e-12)
