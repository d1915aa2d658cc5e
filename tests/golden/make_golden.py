"""Regenerates tests/golden/*.npz -- inputs and expected outputs of the fused convolution.

The reference holds no golden vectors for this path and cannot run here (oracle.c
header: PARITY UNPINNED), so the vectors are produced by this repo's CPU oracle (C, float64
accumulation) and accepted only if the independent torch/autograd restatement
(oracle/torch_ref.py) agrees to 1e-9.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import torch_ref  # noqa: E402


def _csr(src, dst, n):
    d = oracle.coo_to_hyper(src, dst, n)
    return d


def block_batch(rng, sizes, p):
    src, dst, off = [], [], 0
    for n in sizes:
        a = np.triu(rng.random((n, n)) < p, 1)
        i, j = np.nonzero(a | a.T)
        src.append(i + off)
        dst.append(j + off)
        off += n
    return np.concatenate(src), np.concatenate(dst), off


def star(rng, hub_deg, ring):
    """hub 0 -> hub_deg leaves (and back), plus a ring over the leaves."""
    n = hub_deg + 1
    leaves = np.arange(1, n)
    src = [np.zeros(hub_deg, np.int64), leaves]
    dst = [leaves, np.zeros(hub_deg, np.int64)]
    if ring:
        src.append(leaves)
        dst.append(np.roll(leaves, 1))
    return np.concatenate(src), np.concatenate(dst), n


def case(name, src, dst, n, h, f, rng, val=None, isolated=0):
    n_tot = n + isolated  # trailing isolated nodes: empty rows AND empty columns
    d = _csr(src, dst, n_tot)
    nnz = len(src)
    valv = np.ones(nnz, np.float32) if val is None else val.astype(np.float32)
    valv = valv[d["edge_order"]]
    sc = f ** -0.25
    Q, K, V = (rng.standard_normal((n_tot, h, f)).astype(np.float32) * sc for _ in range(3))
    dO = rng.standard_normal((n_tot, h, f)).astype(np.float32)
    arow, acol = (rng.standard_normal((n_tot, h)).astype(np.float32) for _ in range(2))
    out, attn = oracle.gt_forward(d["row_ptr"], d["col_ind"], valv, Q, K, V, want_attn=True)
    dQ, dK, dV = oracle.gt_backward(d["row_ptr"], d["col_ind"], valv, Q, K, V, dO)
    gat = oracle.gat_forward(d["row_ptr"], d["col_ind"], arow, acol, 0.2, V)
    # accept only if the independent restatement agrees
    o2, a2, q2, k2, v2 = torch_ref.gt_forward_backward(d["row_ptr"], d["col_ind"], valv, Q, K, V, dO)
    g2, _ = torch_ref.gat_forward(d["row_ptr"], d["col_ind"], arow, acol, 0.2, V)
    for a, b, what in ((out, o2, "out"), (attn, a2, "attn"), (dQ, q2, "dQ"), (dK, k2, "dK"), (dV, v2, "dV"),
                       (gat, g2, "gat")):
        err = float(np.abs(a - b.numpy()).max()) if a.size else 0.0
        assert err < 1e-9, (name, what, err)
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), num_nodes=np.int64(n_tot), src=src.astype(np.int32),
        dst=dst.astype(np.int32), row_ptr=d["row_ptr"], col_ind=d["col_ind"], rows=d["rows"], val=valv,
        col_ptr=d["col_ptr"], row_ind=d["row_ind"], val_idx=d["val_idx"], Q=Q, K=K, V=V, dO=dO, attn_row=arow,
        attn_col=acol, negative_slope=np.float32(0.2), out=out.astype(np.float32), attn=attn.astype(np.float32),
        dQ=dQ.astype(np.float32), dK=dK.astype(np.float32), dV=dV.astype(np.float32),
        gat_out=gat.astype(np.float32))
    print(f"{name}: m={n_tot} nnz={nnz} h={h} f={f}")


def main():
    rng = np.random.default_rng(20240607)
    s, d, n = block_batch(rng, [23, 40, 31], 0.4)
    case("batch_small_h1_f128", s, d, n, 1, 128, rng)
    s, d, n = block_batch(rng, [17, 29], 0.3)
    case("multihead_h4_f32_isolated", s, d, n, 4, 32, rng, isolated=3)
    s, d, n = star(rng, 200, True)
    case("star200_h1_f64", s, d, n, 1, 64, rng)
    s, d, n = star(rng, 1500, False)
    case("star1500_h1_f16", s, d, n, 1, 16, rng)
    # duplicate edges, self loops, non-unit values
    n = 24
    s = rng.integers(0, n, 160)
    d = rng.integers(0, n, 160)
    s = np.concatenate([s, s[:40], np.arange(6)])
    d = np.concatenate([d, d[:40], np.arange(6)])
    case("dups_selfloops_val_h2_f16", s, d, n, 2, 16, rng, val=rng.uniform(0.25, 2.0, len(s)))
    s, d, n = block_batch(rng, [19, 26], 0.35)
    case("oddf_h3_f20", s, d, n, 3, 20, rng)
    s, d, n = block_batch(rng, [15, 22], 0.35)
    case("oddf_h2_f7", s, d, n, 2, 7, rng)


if __name__ == "__main__":
    main()
