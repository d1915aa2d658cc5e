#!/usr/bin/env python3
"""Secondary measurements: BASELINE.json configs C1 (CPU), C2 (cora-like GAT 'softmax'), C4 (reddit-like GAT
'tiling'), C5 (Peptides-like GT multi-head 'hyper' fwd+bwd).  Prints one JSON line per config.
Timing protocol of the reference: 3 dry + 10 timed calls between device events (DFGNN/utils/util.py:391-400).
"gattrain": the GAT training pair (general CSR / CSC kernels) on the full-graph configs (cora-like, reddit-like).
usage: python3 tests/tools/bench_configs.py [c1] [c2] [c4] [c5] [c3gat] [gattrain] [--reddit-scale S]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from DFGNN.layers import preprocess_Hyper_fw_bw  # noqa: E402
from DFGNN.layers.util import preprocess_CSR, preprocess_softmax  # noqa: E402
from DFGNN.utils import benchmark  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")] or ["c1", "c2", "c4", "c5"]
scale = float(sys.argv[sys.argv.index("--reddit-scale") + 1]) if "--reddit-scale" in sys.argv else 1.0
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
HBM = 8000.0


def emit(**kw):
    print(json.dumps(kw), flush=True)


if "c1" in args:
    import oracle
    g = S.cora_like()
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    ar, ac, X = S.gat_features(g.num_nodes(), 1, 64, seed=4)
    a = [t.numpy() for t in (row_ptr, col_ind, ar, ac, X)]
    for _ in range(3):
        oracle.gat_forward(a[0], a[1], a[2], a[3], 0.2, a[4], acc="f32")
    t0 = time.perf_counter()
    for _ in range(10):
        oracle.gat_forward(a[0], a[1], a[2], a[3], 0.2, a[4], acc="f32")
    dt = (time.perf_counter() - t0) / 10
    emit(config="C1 GAT cora-like f=64 csr, CPU oracle port (fp32, OpenMP)", edges=g.num_edges(), ms=dt * 1e3,
         edges_per_s=g.num_edges() / dt, cores=oracle.num_threads())

if dev == "cpu":
    sys.exit(0)
import oracle  # noqa: E402
from DFGNN.operators import fused_gatconv as gat  # noqa: E402
from DFGNN.operators import fused_gtconv as gt  # noqa: E402

if "c2" in args:
    g = S.cora_like().to(dev)
    row_ptr, col_ind, rows, _, smem = preprocess_softmax(g)
    m, nnz = g.num_nodes(), g.num_edges()
    ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=dev)
    want = oracle.gat_forward(row_ptr.cpu().numpy(), col_ind.cpu().numpy(), ar.cpu().numpy(), ac.cpu().numpy(), 0.2,
                              X.cpu().numpy())
    for name, fn in (("softmax", lambda: gat.GATConvFuse_inference_softmax(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X)),
                     ("softmax_gm", lambda: gat.GATConvFuse_inference_softmax_gm(ar, ac, row_ptr, col_ind, rows, 0.2, X)),
                     ("hyper", lambda: gat.GATConvFuse_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X)),
                     ("tiling", lambda: gat.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X))):
        out, sec = benchmark(fn)
        err = float(np.abs(out.cpu().double().numpy() - want).max())
        byt = 8 * m * 128 + 8 * m + 4 * (m + 1) + 8 * nnz + (8 * nnz if name.startswith("softmax") else 0)
        emit(config=f"C2 GAT cora-like f=128 '{name}'", edges=nnz, us=sec * 1e6, edges_per_s=nnz / sec,
             max_abs_err=err, algorithmic_GBs=byt / sec / 1e9, hbm_frac=byt / sec / 1e9 / HBM)

if "c4" in args:
    t0 = time.perf_counter()
    g = S.reddit_like(scale=scale).to(dev)
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    m, nnz = g.num_nodes(), g.num_edges()
    gen_s = time.perf_counter() - t0
    ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=dev)
    deg = (row_ptr[1:] - row_ptr[:-1])
    out, sec = benchmark(lambda: gat.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X))
    byt = 8 * m * 128 + 8 * m + 4 * (m + 1) + 4 * nnz
    # parity on a row sample (a full oracle run on 114 M edges is too slow for a benchmark run; rows are independent):
    # 200 random rows recomputed in float64 with numpy
    idx = torch.randperm(m, generator=torch.Generator().manual_seed(0))[:200].sort().values
    errs = []
    cp = row_ptr.cpu().numpy()
    for r in idx.numpy():
        cols = col_ind[cp[r]:cp[r + 1]].cpu().numpy()
        s = ar[r, 0].item() + ac[cols, 0].cpu().double().numpy()
        s = np.where(s > 0, s, 0.2 * s)
        p = np.exp(s - s.max()) if len(s) else s
        ref = (p[:, None] * X[cols, 0].cpu().double().numpy()).sum(0) / p.sum() if len(s) else np.zeros(128)
        errs.append(float(np.abs(out[r, 0].cpu().double().numpy() - ref).max()))
    emit(config=f"C4 GAT reddit-like f=128 'tiling' (scale {scale})", nodes=m, edges=nnz, max_degree=int(deg.max()),
         ms=sec * 1e3, edges_per_s=nnz / sec, max_abs_err_200_rows=max(errs), algorithmic_GBs=byt / sec / 1e9,
         hbm_frac=byt / sec / 1e9 / HBM, gather_GBs=nnz * 512 / sec / 1e9, graph_build_s=gen_s)

if "gattrain" in args:
    import fused_gatconv as _gatb
    from DFGNN.layers import preprocess_Hyper_fw_bw as _prep
    graphs = [("cora-like", S.cora_like()), ("Peptides-like bs=256", S.peptides_like(batch_size=256, seed=3))]
    if "--no-reddit" not in sys.argv:
        graphs.append((f"reddit-like (scale {scale})", S.reddit_like(scale=scale)))
    for name, graph in graphs:
        g = graph.to(dev)
        A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = _prep(g)
        m, nnz = g.num_nodes(), g.num_edges()
        ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=dev)
        dO = torch.randn(m, 1, 128, device=dev)
        for drop in (0.0, 0.5):
            torch.manual_seed(0)
            (out, emax, esum, mask), sec_f = benchmark(lambda: _gatb.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, drop))
            (gf, gr, gc), sec_b = benchmark(lambda: _gatb.gat_backward(0.2, drop, row_ptr, col_ind, col_ptr, row_ind,
                                                                       val_idx, emax, esum, mask, X, ar, ac, dO))
            # out is linear in X for fixed attention: <dO, out> == <grad_feat, X>; softmax shift invariance: the row /
            # column gradients of the logits have the same total
            lhs, rhs = float((dO.double() * out.double()).sum()), float((gf.double() * X.double()).sum())
            emit(config=f"GAT training pair on {name}, f=128, attn_drop={drop} (general CSR/CSC kernels)", nodes=m, edges=nnz,
                 fwd_us=sec_f * 1e6, bwd_us=sec_b * 1e6, edges_per_s=nnz / (sec_f + sec_b),
                 linearity_rel_err=abs(lhs - rhs) / max(1e-30, float((dO.double() * out.double()).abs().sum())),
                 grad_row_col_total_diff=abs(float(gr.double().sum()) - float(gc.double().sum())))

if "c3gat" in args:
    import fused_gatconv as _gatb
    g = S.pattern_like(batch_size=1024, seed=1).to(dev)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=dev)
    byt = 8 * m * 128 + 8 * m + 4 * (m + 1) + 8 * nnz
    for use_plan in (True, False):
        _gatb.USE_BLOCK_PLAN = use_plan
        out, sec = benchmark(lambda: gat.GATConvFuse_inference_hyper(smem, ar, ac, row_ptr, col_ind, rows, 0.2, X))
        emit(config=f"GAT 'hyper' on the C3 batch (PATTERN-like bs=1024 f=128) plan={use_plan}", edges=nnz, us=sec * 1e6,
             edges_per_s=nnz / sec, algorithmic_GBs=byt / sec / 1e9, hbm_frac=byt / sec / 1e9 / HBM)
    _gatb.USE_BLOCK_PLAN = True

if "c5" in args:
    import fused_gtconv as _gtb
    from DFGNN.utils import GraphedStep
    runs = []
    for heads, use_plan in ((4, True), (4, False), (8, True), (8, False)):
        _gtb.USE_BLOCK_PLAN = use_plan
        f = 128 // heads
        g = S.peptides_like(batch_size=256, seed=3).to(dev)
        A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
        m, nnz = g.num_nodes(), g.num_edges()
        Q, K, V = (t.requires_grad_(True) for t in S.gt_features(m, heads, f, seed=3, device=dev))
        dO = torch.randn(m, heads, f, device=dev)
        gargs = (rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem)

        def step(gargs=gargs, Q=Q, K=K, V=V, dO=dO):
            out = gt.GTConvFuse_hyper(*gargs, Q, K, V)
            return (out,) + torch.autograd.grad(out, (Q, K, V), dO)

        def raw_step(gargs=gargs, Q=Q, K=K, V=V, dO=dO):   # the same launches as explicit operator calls ...
            rows_, row_ptr_, rest = gargs[0], gargs[1], gargs[2:]
            with torch.no_grad():
                return _gtb.gt_hyper_step_raw(row_ptr_, rest[0], rows_, *rest[1:], Q, K, V, dO)

        # (every eager run before the first capture: a captured graph keeps its memory pool.)  The reference's protocol
        # (3 + 10 calls) five times over, median: one host hiccup inside a 10-call window -- an allocator miss, the
        # collector -- otherwise shows up as a 5x .. 50x outlier of a ~100 us step (seen: 639, 1578 and 5739 us)
        for _ in range(50):      # (the first ~30 steps of a process run at about half speed: clocks / allocator warm-up)
            step()
        trials = [benchmark(step) for _ in range(5)]
        res, sec = trials[0][0], float(np.median([t[1] for t in trials]))
        runs.append((heads, use_plan, f, m, nnz, step, raw_step, res, sec, row_ptr, [round(t[1] * 1e6, 1) for t in trials]))
    for heads, use_plan, f, m, nnz, step, raw_step, res, sec, row_ptr, trials_us in runs:
        _gtb.USE_BLOCK_PLAN = use_plan
        graphed = GraphedStep(raw_step)                      # ... recorded once, replayed as one hipGraphLaunch
        res_g, sec_g = benchmark(graphed.replay)
        assert all(torch.equal(a, b) for a, b in zip(res, res_g))
        gargs = step.__defaults__[0]
        Q, K, V, dO = step.__defaults__[1:]
        n = lambda t: t.detach().cpu().numpy()  # noqa: E731
        rp, ci, vl = n(gargs[1]), n(gargs[2]), n(gargs[3])
        want = oracle.gt_forward(rp, ci, vl, n(Q), n(K), n(V))
        wq, wk, wv = oracle.gt_backward(rp, ci, vl, n(Q), n(K), n(V), n(dO))
        err = max(float(np.abs(n(a).astype(np.float64) - b).max()) for a, b in zip(res, (want, wq, wk, wv)))
        D = 128
        byt = (16 * m * D + 12 * nnz + 4 * (m + 1) + 4 * heads * nnz) + (28 * m * D + 12 * heads * nnz + 16 * nnz + 8 * (m + 1))
        plan = getattr(row_ptr, "_dfgnn_plans", {}).get(f) if use_plan else None  # low-degree batches build none
        emit(config=f"C5 GT Peptides-like bs=256 dim=128 heads={heads} 'hyper' fwd+bwd plan={use_plan}", nodes=m, edges=nnz,
             us=sec * 1e6, us_trials=trials_us, edges_per_s=nnz / sec, us_hipgraph=sec_g * 1e6, edges_per_s_hipgraph=nnz / sec_g,
             max_abs_err=err, algorithmic_GBs=byt / sec / 1e9,
             hbm_frac=byt / sec / 1e9 / HBM, plan_fit=plan.num_fit if plan else 0,
             plan_spill=plan.num_spill if plan else 0)
    _gtb.USE_BLOCK_PLAN = True
