"""Graph-level sharding of a block-diagonal batch and the forward-output all-gather.

Why whole graphs: in a DGL-style batch, graph g owns a contiguous node range and a contiguous CSR edge
range, and every edge stays inside it, so any contiguous run of graphs is a self-contained problem for
the forward AND the backward (dQ/dK/dV of a graph depend only on that graph).  Nothing is exchanged
during fwd+bwd; the only exchange step is the optional all-gather of per-shard outputs for a caller
that needs the whole batch's output on every rank (inference).  Shards are balanced by edge count.

xGMI note: the 8 MI355X of a node are fully connected point to point; RCCL's all-gather moves each
shard once per peer.  Shards have different row counts, so rows are padded to the largest shard for
`all_gather_into_tensor` and trimmed afterwards (one collective, no per-peer send/recv loop).
"""
import torch
import torch.distributed as dist

from DFGNN.utils.graph import Graph


def shard_graph_bounds(batch_num_nodes, edges_per_graph, world_size):
    """Contiguous graph ranges [(g0, g1)] * world_size with near-equal edge counts (greedy prefix cut)."""
    n_graphs = len(batch_num_nodes)
    edges = torch.as_tensor(edges_per_graph, dtype=torch.float64)
    csum = torch.cumsum(edges, 0)
    total = float(csum[-1]) if n_graphs else 0.0
    bounds, g0 = [], 0
    for r in range(world_size):
        if r == world_size - 1:
            g1 = n_graphs
        else:
            target = total * (r + 1) / world_size
            g1 = int(torch.searchsorted(csum, torch.tensor(target, dtype=torch.float64), right=False)) + 1
            g1 = max(g0, min(g1, n_graphs - (world_size - 1 - r)))  # leave >= 1 graph per later rank if possible
            g1 = max(g1, g0 + 1) if g0 < n_graphs else g0
        bounds.append((g0, min(g1, n_graphs)))
        g0 = bounds[-1][1]
    return bounds


def shard_rows(batch_num_nodes, bounds):
    """Node ranges [(n0, n1)] of the graph ranges."""
    off = torch.zeros(len(batch_num_nodes) + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(torch.as_tensor(batch_num_nodes, dtype=torch.int64), 0)
    return [(int(off[g0]), int(off[g1])) for g0, g1 in bounds]


def shard_graph(g, rank, world_size):
    """The sub-batch (whole graphs, node ids rebased to 0) that `rank` owns, plus its node range."""
    sizes = g.batch_num_nodes()
    src, dst = g.edges()
    off = torch.zeros(len(sizes) + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(sizes, 0)
    gid = torch.bucketize(src.cpu(), off[1:], right=True)
    edges_per_graph = torch.bincount(gid, minlength=len(sizes))
    bounds = shard_graph_bounds(sizes, edges_per_graph, world_size)
    (n0, n1) = shard_rows(sizes, bounds)[rank]
    keep = (src >= n0) & (src < n1)
    g0, g1 = bounds[rank]
    sub = Graph(src[keep] - n0, dst[keep] - n0, n1 - n0, sizes[g0:g1])
    for k, v in g.ndata.items():
        sub.ndata[k] = v[n0:n1]
    return sub, (n0, n1)


def all_gather_rows(local, group=None):
    """All-gather tensors that differ in their first dimension; returns the concatenation in rank order."""
    world = dist.get_world_size(group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes)
    send = local
    if local.shape[0] != n_max:
        send = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    else:  # gloo (CPU tests)
        parts = list(recv.chunk(world))
        dist.all_gather(parts, send.contiguous(), group=group)
    return torch.cat([recv[r * n_max: r * n_max + sizes[r]] for r in range(world)])


class ShardedGTConv:
    """Forward of the fused GT conv on this rank's shard, optionally all-gathered.

    conv_fn(params, Q, K, V) -> out[m_local, h, f]; the default is the fused 'hyper' operator.  The
    gather runs on the current stream after the kernel (RCCL orders it); pass gather=False for the
    training step, where nothing needs to be exchanged."""

    def __init__(self, conv_fn=None, group=None):
        if conv_fn is None:
            from DFGNN.operators.fused_gtconv import GTConvFuse_inference_hyper

            def conv_fn(params, Q, K, V):
                indptr, indices, rows, val, smem = params
                return GTConvFuse_inference_hyper(indptr, indices, rows, val, smem, Q, K, V)
        self.conv_fn, self.group = conv_fn, group

    def __call__(self, params, Q, K, V, gather=True):
        out = self.conv_fn(params, Q, K, V)
        return all_gather_rows(out, self.group) if gather else out
