"""CPU: host-side logic -- preprocessing vs scipy, binding checks, baseline (fuse=False) layer
branches vs the oracle, synthetic generators, and that there is no silent CPU fallback."""
import argparse

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import DFGNN  # noqa: F401
from DFGNN.layers import load_graphconv_layer, load_prepfunc, preprocess_Hyper_fw_bw
from DFGNN.layers.util import preprocess_CSR, preprocess_Hyper, preprocess_softmax
from DFGNN.utils import Graph, batch, check_correct, preprocess_dglsp
from DFGNN.utils import synthetic as S


def test_preprocess_matches_scipy_and_oracle(oracle_mod):
    g = S.pattern_like(batch_size=6, seed=11)
    src, dst = (t.numpy() for t in g.edges())
    n = g.num_nodes()
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    ref = oracle_mod.coo_to_hyper(src, dst, n)
    for k, t in dict(rows=rows, row_ptr=row_ptr, col_ind=col_ind, col_ptr=col_ptr, row_ind=row_ind,
                     val_idx=val_idx).items():
        assert t.dtype == torch.int32
        np.testing.assert_array_equal(t.numpy(), ref[k], err_msg=k)
    M = sp.coo_matrix((np.ones(len(src)), (src, dst)), shape=(n, n))
    csr, csc = M.tocsr(), M.tocsc()
    np.testing.assert_array_equal(row_ptr.numpy(), csr.indptr)
    np.testing.assert_array_equal(col_ptr.numpy(), csc.indptr)
    # val_idx maps CSC slots to CSR slots of the same (row, col) entry
    r_of = np.repeat(np.arange(n), np.diff(row_ptr.numpy()))
    np.testing.assert_array_equal(r_of[val_idx.numpy()], row_ind.numpy())
    c_of = np.repeat(np.arange(n), np.diff(col_ptr.numpy()))
    np.testing.assert_array_equal(col_ind.numpy()[val_idx.numpy()], c_of)
    assert smem == 1024 and val.dtype == torch.float32 and bool((val == 1).all())
    assert len(preprocess_CSR(g)) == 4 and len(preprocess_Hyper(g)) == 5 and len(preprocess_softmax(g)) == 5
    assert preprocess_Hyper_fw_bw(g, fused=False)[1] is None


def test_batch_is_block_diagonal():
    gs = [S.pattern_like(batch_size=1, seed=s) for s in range(3)]
    b = batch(gs)
    assert b.num_nodes() == sum(g.num_nodes() for g in gs)
    s, d = b.edges()
    bounds = torch.cumsum(b.batch_num_nodes(), 0)
    gid = lambda x: torch.bucketize(x, bounds, right=True)  # noqa: E731
    assert bool((gid(s) == gid(d)).all())


def test_generators_shapes():
    g = S.cora_like()
    assert g.num_nodes() == 2708 and g.num_edges() == 10556
    s, d = g.edges()
    assert bool((s != d).all())
    g = S.peptides_like(batch_size=8)
    deg = torch.bincount(g.edges()[0], minlength=g.num_nodes())
    assert int(deg.max()) <= 5 and int((deg == 0).sum()) >= 1
    g = S.reddit_like(scale=0.002)
    assert g.num_edges() % 2 == 0 and g.num_nodes() >= 64


def test_binding_rejects_cpu_tensors_and_bad_dtypes():
    """Error behaviour of the reference binding: RuntimeError '<x> must be on CUDA'
    (fused_gtconv.cpp:7-13).  There is no CPU fallback."""
    import fused_gatconv
    import fused_gtconv
    ip = torch.tensor([0, 1], dtype=torch.int32)
    idx = torch.tensor([0], dtype=torch.int32)
    val = torch.ones(1)
    q = torch.ones(1, 1, 4)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fused_gtconv.gt_hyper_inference(ip, idx, idx, val, 1024, q, q, q)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fused_gtconv.gt_tiling_inference(ip, idx, val, 128, q, q, q)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fused_gatconv.gat_inference_tiling(torch.ones(1, 1), torch.ones(1, 1), ip, idx, 0.2, q)
    a = torch.ones(1, 1)
    with pytest.raises(RuntimeError, match="must be on CUDA"):       # training pair: no CPU fallback either
        fused_gatconv.gat_forward(a, a, ip, idx, 0.2, q, 0.0)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fused_gatconv.gat_backward(0.2, 0.0, ip, idx, ip, idx, idx, a, a, a, q, a, a, q)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fused_gatconv.gat_inference_hyper_v2(1024, a, a, ip, idx, 0.2, q)
    with pytest.raises(RuntimeError, match="must be on CUDA"):
        fused_gatconv.gat_inference_hyper_recompute(a, a, ip, idx, 0.2, q)
    with pytest.raises(RuntimeError, match="must be on CUDA"):       # the tile-scheduler entry point: same forward
        fused_gatconv.gat_forward_tb(a, a, ip, idx, 0.2, q, torch.zeros(1, 2, dtype=torch.int32))


def _args(conv, fmt, dim, heads):
    return argparse.Namespace(conv=conv, format=fmt, dim=dim, heads=heads)


@pytest.mark.parametrize("conv,fmt", [("gt", "hyper"), ("gt", "softmax"), ("gt", "tiling"), ("gt", "softmax_gm"),
                                      ("gat", "hyper"), ("gat", "softmax"), ("gat", "tiling"),
                                      ("gat", "softmax_gm"), ("gat", "hyper_v2"), ("gat", "hyper_recompute"),
                                      ("gat", "hyper_ablation"), ("gt", "hyper_ablation"), ("agnn", "hyper"),
                                      ("agnn", "csr"), ("agnn", "softmax_gm"), ("agnn", "csr_gm")])
def test_layer_baseline_branch_matches_oracle(oracle_mod, conv, fmt):
    """fuse=False branch (torch restatement of dgl.sparse) at the layer boundary vs the oracle at the
    operator boundary, through the layer's own layout transforms."""
    torch.manual_seed(0)
    dim, heads = 32, 4
    if conv == "agnn":
        heads = 1  # the reference's AGNN projects to out_size, not out_size * heads (agnn_layer.py:12): single head
    layer = load_graphconv_layer(_args(conv, fmt, dim, heads)).eval()
    assert load_prepfunc(_args(conv, fmt, dim, heads)) is not None
    g = S.pattern_like(batch_size=3, seed=5)
    n = g.num_nodes()
    A = preprocess_dglsp(g)
    x = torch.randn(n, dim)
    with torch.no_grad():
        out, ms = layer(A, x, fuse=False)
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    if conv == "agnn":
        Hb = layer.proj(x).view(-1, layer.out_size, heads).detach()          # baseline layout [N, out, heads]
        Ho = Hb.transpose(1, 2).contiguous()                                 # operator layout [N, heads, out]
        Hn = torch.nn.functional.normalize(Ho, p=2, dim=-1)
        ref = oracle_mod.gt_forward(row_ptr.numpy(), col_ind.numpy(), val.numpy(), Hn.numpy(), Hn.numpy(), Ho.numpy())
        ref = torch.from_numpy(ref).reshape(n, -1)
        assert out.shape == (n, dim * heads)
    elif conv == "gt":
        q, k, v = layer.prep_qkv(x)
        q, k, v = (t.transpose(1, 2).contiguous().detach().numpy() for t in (q, k, v))
        ref = oracle_mod.gt_forward(row_ptr.numpy(), col_ind.numpy(), val.numpy(), q, k, v)
        ref = torch.from_numpy(ref).transpose(1, 2).reshape(n, -1)
        assert out.shape == (n, dim)
    else:
        z = layer.W(x).view(-1, layer.out_size, heads).detach()              # baseline layout [N, out, heads]
        zt = z.transpose(1, 2).contiguous()                                 # operator layout [N, heads, out]
        a_l, a_r = layer.a_l.transpose(1, 2).detach(), layer.a_r.transpose(1, 2).detach()
        ar, ac = (a_l * zt).sum(-1), (a_r * zt).sum(-1)
        ref = oracle_mod.gat_forward(row_ptr.numpy(), col_ind.numpy(), ar.numpy(), ac.numpy(), 0.2, zt.numpy())
        ref = torch.from_numpy(ref).transpose(1, 2).reshape(n, -1)
        assert out.shape == (n, dim * heads)
    assert torch.allclose(out.double(), ref, atol=1e-4)
    assert check_correct(out, ref.float())


def test_layer_factory_errors():
    with pytest.raises(ValueError):
        load_graphconv_layer(_args("gt", "cugraph", 8, 1))
    with pytest.raises(ValueError):
        load_graphconv_layer(_args("sage", "hyper", 8, 1))


def test_loader_needs_no_import_order():
    """dfgnn_native.lib() in a fresh interpreter that has not imported torch: libdfgnn.so is a forwarder without a HIP
    dependency (csrc/gen_shim.py), so nothing has to be imported first and loading it brings no HIP runtime in (round 2's
    loader had to import torch ahead of the library; tests/test_capi_symbols.py has the details)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = ("import sys; sys.path.insert(0, %r); import dfgnn_native as n; L = n.lib(); assert 'torch' not in sys.modules; "
            "assert L.dfgnn_abi_version() > 0 and 'libamdhip64' not in open('/proc/self/maps').read(); print('ok')"
            % (ROOT + "/df-gnn_amd"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_synthetic_datasets_and_loader():
    """DFGNN.utils.datasets: the stand-ins for load_dataset_fn / load_data_full_graph / GraphDataLoader behave like the
    objects the reference's scripts expect (block-diagonal batches, ndata['feat'], deterministic items)."""
    from DFGNN.utils import GraphDataLoader, load_data_full_graph, load_dataset_fn
    ds, infer = load_dataset_fn("PATTERN", None, length=7)
    assert len(ds) == 7 and callable(infer)
    g0, y0 = ds[0]
    g0b, _ = ds[0]
    assert torch.equal(g0.edges()[0], g0b.edges()[0]) and torch.equal(g0.ndata["feat"], g0b.ndata["feat"])
    batches = list(GraphDataLoader(ds, batch_size=3))
    assert len(batches) == 3 and [len(y) for _, y in batches] == [3, 3, 1]
    bg, _ = batches[0]
    sizes = [ds[i][0].num_nodes() for i in range(3)]
    assert bg.num_nodes() == sum(sizes) and bg.ndata["feat"].shape == (sum(sizes), 64)
    src, dst = bg.edges()
    off = np.cumsum([0] + sizes)
    gid = np.searchsorted(off, src.numpy(), side="right")
    assert (gid == np.searchsorted(off, dst.numpy(), side="right")).all()      # no edge crosses a member graph
    with pytest.raises(ValueError):
        load_dataset_fn("ogbg-molhiv", None)
    cora = load_data_full_graph("cora")
    assert cora.num_nodes() == 2708 and cora.ndata["feat"].shape[0] == 2708
    with pytest.raises(NotImplementedError):
        GraphDataLoader(ds, batch_size=2, shuffle=True)


def test_check_correct_follows_the_reference_rule(capsys):
    """isclose(rtol=1e-3) per element, a row with exactly one miss passes, every failing row is examined
    (DFGNN/utils/util.py:211-236 of the reference); the check is relative only (atol 1e-8)."""
    a = torch.randn(6, 8)
    b = a.clone()
    assert check_correct(a, b)
    b[1, 2] += 1.0
    b[4, 0] += 1.0                                   # two rows with a single miss each: tolerated
    assert check_correct(a, b)
    b[4, 5] += 1.0                                   # second miss in row 4
    assert not check_correct(a, b)
    assert "error node 4 mismatch" in capsys.readouterr().out
    c = torch.zeros(3, 4)
    c[:, 0] = 1.0
    d = c.clone()
    d[2, 1] = 3e-6
    d[2, 2] = -2e-6                                  # near-zero elements: relative check trips, absolute error tiny
    assert not check_correct(c, d)
    assert "error node 2 mismatch" in capsys.readouterr().out


def test_dotgat_baseline_restates_dgl_dotgatconv():
    """DotGatConv (the torch restatement of dgl.nn.DotGatConv behind the DOTGAT layers' non-fused branch) against a
    per-node loop of the published algorithm: e_uv = <h_u, h_v> / sqrt(out_feats), softmax over the edges arriving at v,
    sum of a_uv h_u; on a symmetric graph it equals the GT oracle with Q = H / sqrt(out_feats), K = V = H (what the fused
    branch computes).  dgl itself is absent: parity with the module it restates is unpinned."""
    import oracle
    from DFGNN.layers.GAT_DOT import DotGatConv
    from DFGNN.layers.util import preprocess_CSR
    from DFGNN.utils import synthetic as S
    torch.manual_seed(0)
    g = S.pattern_like(batch_size=3, seed=4)
    conv = DotGatConv(12, 5, 2)
    x = torch.randn(g.num_nodes(), 12) * 0.3
    with torch.no_grad():
        out = conv(g, x)
        h = conv.fc(x).view(-1, 2, 5)
    src, dst = g.edges()
    want = torch.zeros_like(out)
    for v in range(g.num_nodes()):
        us = src[dst == v]
        if len(us) == 0:
            continue
        for hd in range(2):
            e = (h[us, hd] * h[v, hd]).sum(-1) / 5 ** 0.5
            a = torch.softmax(e, 0)
            want[v, hd] = (a[:, None] * h[us, hd]).sum(0)
    assert torch.allclose(out, want, atol=1e-5)
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    hn = h.numpy()
    ref = oracle.gt_forward(row_ptr.numpy(), col_ind.numpy(), val.numpy(), (hn / np.float32(5 ** 0.5)), hn, hn)
    assert np.abs(out.numpy() - ref).max() < 1e-5


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts N rank processes itself --
    before anything touches a GPU -- and rank 0's line says n_gpus = N; a launcher that started another number of ranks
    than --gpus is refused.  (--dry-run: rendezvous and collectives over gloo, no kernels; the GPU suite runs the real
    line.)"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] is True and line["steps"] == 3
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert one.returncode == 0 and json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "refusing" in bad.stderr
