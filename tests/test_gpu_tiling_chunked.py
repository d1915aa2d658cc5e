"""GPU parity tests of the chunked GAT 'tiling' (csrc/gat_tiling_chunked.hip; include/dfgnn.h:
dfgnn_gat_tiling_chunked_fwd): column chunks per XCD, per-(row, chunk) online-softmax partial states, merge pass --
against the CPU oracle and the single-kernel form, through the binding -> C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, what, atol=1e-3, rtol=1e-3):
    got = got.detach().cpu().double().numpy()
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    assert not (err > atol + rtol * np.abs(want)).any(), f"{what}: max abs err {err.max():.3e}"


@pytest.mark.parametrize("h,f,chunk_rows", [(1, 128, 64), (2, 32, 100), (1, 20, 33), (3, 64, 4096)])
def test_chunked_tiling_matches_oracle_and_single_kernel(oracle_mod, monkeypatch, h, f, chunk_rows):
    """A heavy-tailed graph (hubs of several hundred edges, isolated nodes, rows whose edges all fall into one chunk, chunk
    sizes that do not divide m) with the chunked form forced on."""
    import fused_gatconv as gat
    from DFGNN.layers import preprocess_CSR
    from DFGNN.utils import Graph
    from DFGNN.utils import synthetic as S
    g0 = S.reddit_like(scale=0.004)                       # ~930 nodes, ~460 k edges
    s, d = g0.edges()
    m = g0.num_nodes() + 7                                 # 7 isolated nodes at the end (empty rows, an empty last chunk part)
    g = Graph(s.numpy(), d.numpy(), m).to(DEV)
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    ar, ac, X = S.gat_features(m, h, f, seed=4, device=DEV)
    want = oracle_mod.gat_forward(row_ptr.cpu().numpy(), col_ind.cpu().numpy(), ar.cpu().numpy(), ac.cpu().numpy(), 0.2,
                                  X.cpu().numpy())
    monkeypatch.setattr(gat, "TILING_CHUNK_ROWS", 0)
    plain = gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)
    monkeypatch.setattr(gat, "TILING_CHUNK_ROWS", chunk_rows)
    monkeypatch.setattr(gat, "TILING_CHUNK_MIN_TABLE", 0)
    monkeypatch.setattr(gat, "TILING_CHUNK_MIN_DEGREE", 0)
    assert gat._use_chunked_tiling(m, col_ind.numel(), h, f)
    chunked = gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)
    again = gat.gat_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)     # (cached chunk structure, queues re-zeroed)
    _close(chunked, want, "chunked tiling")
    assert torch.equal(chunked, again)                                       # deterministic: fixed merge order
    assert torch.allclose(chunked, plain, atol=1e-5, rtol=1e-4)
    assert (chunked[-7:] == 0).all()                                         # empty rows -> 0
    # the chunk structure: every edge once, rows in order inside a chunk, columns inside their chunk
    seg_ptr, ccol = gat._tiling_chunks(row_ptr, col_ind, chunk_rows)
    nchunks = (m + chunk_rows - 1) // chunk_rows
    sp = seg_ptr.cpu().numpy()
    assert sp[0] == 0 and sp[-1] == col_ind.numel() and (np.diff(sp) >= 0).all() and len(sp) == nchunks * m + 1
    cc = ccol.cpu().numpy()
    assert cc.min() >= 0 and cc.max() < chunk_rows
    rp, ci = row_ptr.cpu().numpy(), col_ind.cpu().numpy()
    r = 17
    mine = sorted(ci[rp[r]:rp[r + 1]])
    back = sorted(int(c * chunk_rows + x) for c in range(nchunks) for x in cc[sp[c * m + r]:sp[c * m + r + 1]])
    assert mine == back


def test_chunked_tiling_argument_errors():
    import dfgnn_native
    L = dfgnn_native.lib()
    assert L.dfgnn_gat_tiling_chunked_ws_bytes(0, 1, 128, 8192) == 0
    assert L.dfgnn_gat_tiling_chunked_ws_bytes(100, 1, 128, 64) >= 256 + 2 * 100 * 4 * 130
    x = torch.zeros(16, device=DEV)
    P = lambda t: t.data_ptr()  # noqa: E731
    assert L.dfgnn_gat_tiling_chunked_fwd(4, 0, 1, 4, 0, P(x), None, P(x), P(x), 0.2, P(x), P(x), P(x), 64, None) == -1  # chunk_rows
    assert L.dfgnn_gat_tiling_chunked_fwd(4, 0, 1, 4, 2, P(x), None, P(x), P(x), 0.2, P(x), P(x), P(x), 8, None) == -1  # workspace
