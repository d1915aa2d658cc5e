"""Regenerates tests/golden/gat_train/*.npz -- expected outputs of the GAT training pair (gat_forward /
gat_backward behind FusedGATFunction, DFGNN/operators/fused_gatconv.py:95-176) on the inputs of the committed
fixtures tests/golden/*.npz (row_ptr, col_ind, attn_row, attn_col, V as in_feat, dO as the output gradient).

Each file holds, for the fixture of the same name: the attention-dropout randoms ``edge_mask[nnz, h]`` (seeded
here; the reference draws them with cuRAND seeded by clock(), so no reference run could pin them) with
``attn_drop``, and the expected ``out / edge_max / edge_sum / grad_feat / grad_attn_row / grad_attn_col`` with
dropout (``*_drop``) and without.  Same acceptance rule as make_golden.py: the C oracle (float64) must agree
with the torch/autograd restatement to 1e-9 -- PARITY UNPINNED, the reference holds no vectors for this path.
Run from the repo root:  python tests/golden/make_golden_gat_train.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from oracle import torch_ref  # noqa: E402

ATTN_DROP = 0.25


def main():
    rng = np.random.default_rng(20241004)
    for fn in sorted(os.listdir(HERE)):
        if not fn.endswith(".npz"):
            continue
        g = dict(np.load(os.path.join(HERE, fn), allow_pickle=False))
        ip, idx, ar, ac, X, dO = g["row_ptr"], g["col_ind"], g["attn_row"], g["attn_col"], g["V"], g["dO"]
        slope = float(g["negative_slope"])
        h = X.shape[1]
        mask = rng.random((len(idx), h)).astype(np.float32)
        save = dict(edge_mask=mask, attn_drop=np.float32(ATTN_DROP))
        for sfx, mk, drop in (("", None, 0.0), ("_drop", mask, ATTN_DROP)):
            out, emax, esum = oracle.gat_train_forward(ip, idx, ar, ac, slope, X, mk, drop)
            gf, gr, gc = oracle.gat_backward(ip, idx, ar, ac, slope, X, dO, mk, drop)
            o2, gf2, gr2, gc2 = torch_ref.gat_train(ip, idx, ar, ac, slope, X, dO, mk, drop)
            for a, b, what in ((out, o2, "out"), (gf, gf2, "grad_feat"), (gr, gr2, "grad_row"), (gc, gc2, "grad_col")):
                err = float(np.abs(a - b.numpy()).max()) if a.size else 0.0
                assert err < 1e-9, (fn, sfx, what, err)
            if not sfx:  # without dropout the forward equals the inference fixture
                assert float(np.abs(out - g["gat_out"]).max()) < 2e-6
            for k, v in (("out", out), ("edge_max", emax), ("edge_sum", esum), ("grad_feat", gf),
                         ("grad_attn_row", gr), ("grad_attn_col", gc)):
                save[k + sfx] = v.astype(np.float32)
        np.savez_compressed(os.path.join(HERE, "gat_train", fn), **save)
        print(fn, "nnz", len(idx), "kept", float((mask > ATTN_DROP).mean()))


if __name__ == "__main__":
    main()
