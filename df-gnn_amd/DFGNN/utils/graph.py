"""Minimal graph container standing in for the DGLGraph the reference's harness passes around.

Only what the hot path's callers touch is provided: `edges()`, `num_nodes()`, `num_edges()`,
`ndata`, `to(device)` (DFGNN/layers/util.py:52-57, DFGNN/utils/util.py:239-243 in the reference),
plus `batch()` which, like `dgl.batch`, lays graphs out block-diagonally: graph g owns a
contiguous node range and all of its edges stay inside it.
A real DGLGraph duck-types to the same calls, so the preprocess functions accept either.
"""
import torch


class Graph:
    def __init__(self, src, dst, num_nodes, batch_num_nodes=None):
        self._src = torch.as_tensor(src, dtype=torch.int64)
        self._dst = torch.as_tensor(dst, dtype=torch.int64)
        self._n = int(num_nodes)
        self.ndata = {}
        # nodes per member graph (None for a single graph) -- same role as DGLGraph.batch_num_nodes()
        self._batch_num_nodes = None if batch_num_nodes is None else torch.as_tensor(batch_num_nodes, dtype=torch.int64)

    def edges(self):
        return self._src, self._dst

    def num_nodes(self):
        return self._n

    def num_edges(self):
        return int(self._src.numel())

    def batch_num_nodes(self):
        if self._batch_num_nodes is None:
            return torch.tensor([self._n], dtype=torch.int64)
        return self._batch_num_nodes

    @property
    def device(self):
        return self._src.device

    def to(self, device):
        g = Graph(self._src.to(device), self._dst.to(device), self._n, self._batch_num_nodes)
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        return g


def batch(graphs):
    """Block-diagonal union of graphs (node ids of graph g shifted by the nodes before it)."""
    srcs, dsts, sizes, off = [], [], [], 0
    for g in graphs:
        s, d = g.edges()
        srcs.append(s + off)
        dsts.append(d + off)
        sizes.append(g.num_nodes())
        off += g.num_nodes()
    out = Graph(torch.cat(srcs) if srcs else torch.zeros(0, dtype=torch.int64),
                torch.cat(dsts) if dsts else torch.zeros(0, dtype=torch.int64), off, sizes)
    keys = set.intersection(*[set(g.ndata) for g in graphs]) if graphs else set()
    for k in keys:
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs])
    return out
