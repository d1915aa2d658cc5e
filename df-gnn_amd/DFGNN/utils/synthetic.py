"""Synthetic graphs shaped like the datasets BASELINE.json names (no datasets are available
offline).  Statistics come from the titles of the reference's figure/graph_statistics/*.png
(SURVEY.md 6, 8d); reddit's size is external knowledge (DGL RedditDataset).

  cora_like()       C1/C2  m=2 708, nnz=10 556, heavy-tailed degrees (mean 3.9, max 168), symmetric
  pattern_like()    C3     bs graphs, nodes ~ clip(N(118.91, 21.07), 50, 186), symmetric G(n, p) with
                           p = 51.13/117.91 so the mean degree is 51.13; block-diagonal batch
  reddit_like()     C4     m=232 965, nnz ~ 114.6 M, Chung-Lu with Zipf weights (mean ~492, max ~21.6 k)
  peptides_like()   C5     bs graphs, nodes ~ clip(N(150.94, 84.24), 8, 444), forests with degree <= 5,
                           mean ~2, at least one isolated node per batch
All generators are deterministic in `seed` and return a DFGNN.utils.Graph (edge list row=src,
col=dst as in DFGNN/layers/util.py:52-57 of the reference).
"""
import numpy as np
import torch

from .graph import Graph


def _sym(u, v):
    return np.concatenate([u, v]), np.concatenate([v, u])


def pattern_like(batch_size=1024, seed=1, mean_nodes=118.91, std_nodes=21.07, lo=50, hi=186, mean_deg=51.13):
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.rint(rng.normal(mean_nodes, std_nodes, batch_size)), lo, hi).astype(np.int64)
    p = mean_deg / (mean_nodes - 1.0)
    srcs, dsts, off = [], [], 0
    for n in sizes:
        iu, ju = np.triu_indices(int(n), k=1)
        keep = rng.random(iu.shape[0]) < p
        s, d = _sym(iu[keep], ju[keep])
        srcs.append(s + off)
        dsts.append(d + off)
        off += int(n)
    return Graph(np.concatenate(srcs), np.concatenate(dsts), off, sizes)


def peptides_like(batch_size=256, seed=3, mean_nodes=150.94, std_nodes=84.24, lo=8, hi=444):
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.rint(rng.normal(mean_nodes, std_nodes, batch_size)), lo, hi).astype(np.int64)
    srcs, dsts, off = [], [], 0
    for n in sizes:
        n = int(n)
        deg = np.zeros(n, dtype=np.int64)
        u_list, v_list = [], []
        for i in range(1, n):
            if rng.random() < 0.02:          # start a new component: node i may stay isolated
                continue
            lo_p = max(0, i - 4)
            cand = [j for j in range(lo_p, i) if deg[j] < 4]
            if not cand:
                continue
            j = cand[int(rng.integers(len(cand)))]
            u_list.append(j)
            v_list.append(i)
            deg[j] += 1
            deg[i] += 1
        if n >= 3 and u_list:               # guarantee an isolated node in every graph of >= 3 nodes
            last = n - 1
            keep = [(a, b) for a, b in zip(u_list, v_list) if a != last and b != last]
            u_list, v_list = [a for a, _ in keep], [b for _, b in keep]
        s, d = _sym(np.asarray(u_list, dtype=np.int64), np.asarray(v_list, dtype=np.int64))
        srcs.append(s + off)
        dsts.append(d + off)
        off += n
    return Graph(np.concatenate(srcs), np.concatenate(dsts), off, sizes)


def _chung_lu_undirected(m, n_undirected, weights, rng, chunk=1 << 24):
    """Sample undirected pairs with endpoint probability proportional to `weights`; self loops and
    duplicate pairs are dropped, sampling continues until exactly n_undirected unique pairs exist."""
    cdf = np.cumsum(weights / weights.sum())
    cdf[-1] = 1.0
    keys = np.zeros(0, dtype=np.int64)
    while keys.shape[0] < n_undirected:
        need = n_undirected - keys.shape[0]
        k = int(min(chunk, max(1024, need * 1.3)))
        a = np.searchsorted(cdf, rng.random(k)).astype(np.int64)
        b = np.searchsorted(cdf, rng.random(k)).astype(np.int64)
        ok = a != b
        lo, hi = np.minimum(a[ok], b[ok]), np.maximum(a[ok], b[ok])
        keys = np.unique(np.concatenate([keys, lo * m + hi]))
        if keys.shape[0] > n_undirected:
            keys = rng.permutation(keys)[:n_undirected]
    return keys // m, keys % m


def cora_like(seed=0, m=2708, nnz=10556, max_deg=168):
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, m + 1) ** 0.59          # heavy tail: top node ~ max_deg, mean 3.9
    w = rng.permutation(w)
    u, v = _chung_lu_undirected(m, nnz // 2, w, rng)
    s, d = _sym(u, v)
    return Graph(s, d, m)


def reddit_like(seed=2, m=232965, nnz=114_615_892, scale=1.0):
    """`scale` < 1 shrinks both nodes and edges (tests); scale=1 is the full C4 shape."""
    rng = np.random.default_rng(seed)
    m = max(64, int(m * scale))
    n_und = max(64, int(nnz * scale) // 2)
    n_und = min(n_und, m * (m - 1) // 4)        # keep the dedup sampler far from the complete graph
    w = 1.0 / np.arange(1, m + 1) ** 0.34          # Zipf-like: max degree ~ 21.6k at full scale
    w = rng.permutation(w)
    u, v = _chung_lu_undirected(m, n_und, w, rng)
    s, d = _sym(u, v)
    return Graph(s, d, m)


def gt_features(m, heads, dim_per_head, seed, device="cpu"):
    """Q, K, V ~ N(0,1) * f^-0.25 so logits are O(1) like scaled dot-product attention."""
    g = torch.Generator().manual_seed(seed)
    sc = float(dim_per_head) ** -0.25
    return tuple((torch.randn(m, heads, dim_per_head, generator=g) * sc).to(device) for _ in range(3))


def gat_features(m, heads, dim_per_head, seed, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    attn_row = torch.randn(m, heads, generator=g).to(device)
    attn_col = torch.randn(m, heads, generator=g).to(device)
    x = torch.randn(m, heads, dim_per_head, generator=g).to(device)
    return attn_row, attn_col, x
