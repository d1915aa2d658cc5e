"""CPU, world_size 2, gloo: graph-level sharding + output all-gather (the N > 1 path of bench.py).
The local compute is injected (the oracle stands in for the HIP operator, which needs a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from DFGNN.parallel import all_gather_rows, shard_graph, shard_graph_bounds, shard_rows
from DFGNN.utils import synthetic as S


def test_shard_bounds_balanced_and_contiguous():
    g = S.pattern_like(batch_size=64, seed=2)
    sizes = g.batch_num_nodes()
    src, _ = g.edges()
    off = torch.zeros(len(sizes) + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(sizes, 0)
    epg = torch.bincount(torch.bucketize(src, off[1:], right=True), minlength=len(sizes))
    for world in (1, 2, 4, 8):
        b = shard_graph_bounds(sizes, epg, world)
        assert b[0][0] == 0 and b[-1][1] == len(sizes) and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        loads = [int(epg[g0:g1].sum()) for g0, g1 in b]
        assert max(loads) <= 1.35 * (sum(loads) / world)
        rows = shard_rows(sizes, b)
        assert rows[-1][1] == g.num_nodes()
    # shards are self-contained: no edge leaves its shard
    for rank in range(4):
        sub, (n0, n1) = shard_graph(g, rank, 4)
        s, d = sub.edges()
        assert int(s.min()) >= 0 and int(d.min()) >= 0 and int(s.max()) < n1 - n0 and int(d.max()) < n1 - n0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from DFGNN.layers.util import preprocess_Hyper
        from DFGNN.parallel import ShardedGTConv
        g = S.pattern_like(batch_size=12, seed=5)
        Q, K, V = S.gt_features(g.num_nodes(), 2, 16, seed=9)
        sub, (n0, n1) = shard_graph(g, rank, world)

        def conv(params, q, k, v):  # oracle stands in for the HIP op on CPU
            indptr, indices, rows, val, smem = params
            o = oracle.gt_forward(indptr.numpy(), indices.numpy(), val.numpy(), q.numpy(), k.numpy(), v.numpy())
            return torch.from_numpy(o).float()

        op = ShardedGTConv(conv_fn=conv)
        full = op(preprocess_Hyper(sub), Q[n0:n1], K[n0:n1], V[n0:n1], gather=True)
        local = op(preprocess_Hyper(sub), Q[n0:n1], K[n0:n1], V[n0:n1], gather=False)
        assert local.shape[0] == n1 - n0
        ip, idx, rows, val, _ = preprocess_Hyper(g)
        want = oracle.gt_forward(ip.numpy(), idx.numpy(), val.numpy(), Q.numpy(), K.numpy(), V.numpy())
        err = float(np.abs(full.numpy() - want).max())
        # uneven first dimensions through all_gather_rows
        t = torch.full((rank + 1, 3), float(rank))
        cat = all_gather_rows(t)
        ok = cat.shape == (sum(range(1, world + 1)), 3) and bool((cat[0] == 0).all()) and bool((cat[-1] == world - 1).all())
        ret[rank] = (err, full.shape[0], ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_two_rank_shard_and_allgather_matches_single_process(oracle_mod):
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    g = S.pattern_like(batch_size=12, seed=5)
    for rank in range(world):
        err, rows, ok = ret[rank]
        assert err < 1e-6 and rows == g.num_nodes() and ok


def _train_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        from train_stack import Stack
        from DFGNN.layers import preprocess_Hyper_fw_bw
        from DFGNN.parallel import OverlappedGradSync
        torch.manual_seed(0)
        model = Stack(16, 3)
        g = S.pattern_like(batch_size=6, seed=7)
        x = torch.randn(g.num_nodes(), 16, generator=torch.Generator().manual_seed(1))
        y = torch.randn(g.num_nodes(), 1, generator=torch.Generator().manual_seed(2))
        sub, (n0, n1) = shard_graph(g, rank, world)
        grads = {}
        # (hooks accumulate: one sync object at a time is registered on a fresh copy of the model)
        import copy
        models = {True: model, False: copy.deepcopy(model)}
        syncs = {o: OverlappedGradSync(models[o], overlap=o) for o in (True, False)}
        for overlap in (True, False):
            model = models[overlap]
            model.zero_grad(set_to_none=True)
            sync = syncs[overlap]
            params = (preprocess_Hyper_fw_bw(sub, fused=False)[0],) + (None,) * 8
            loss = ((model(params, x[n0:n1], False) - y[n0:n1]) ** 2).sum() / g.num_nodes()
            loss.backward()
            sync.finish()
            grads[overlap] = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
        ret[rank] = (grads[True], grads[False])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_overlapped_weight_gradient_exchange_matches_full_batch():
    """Data-parallel training step over graph shards (world 2, gloo, non-fused torch branch on CPU): the weight gradients
    exchanged bucket by bucket DURING the backward (DFGNN/parallel/overlap.py) equal those exchanged after it, and both
    equal the single-process gradient of the whole batch."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from train_stack import Stack
    from DFGNN.layers import preprocess_Hyper_fw_bw
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_train_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    torch.manual_seed(0)
    model = Stack(16, 3)
    g = S.pattern_like(batch_size=6, seed=7)
    x = torch.randn(g.num_nodes(), 16, generator=torch.Generator().manual_seed(1))
    y = torch.randn(g.num_nodes(), 1, generator=torch.Generator().manual_seed(2))
    params = (preprocess_Hyper_fw_bw(g, fused=False)[0],) + (None,) * 8
    (((model(params, x, False) - y) ** 2).sum() / g.num_nodes()).backward()
    want = torch.cat([p.grad.flatten() for p in model.parameters()])
    for rank in range(world):
        g_overlap, g_plain = ret[rank]
        assert torch.allclose(g_overlap, g_plain, atol=1e-7)
        assert torch.allclose(g_overlap, want, atol=1e-6, rtol=1e-5)


def _partial_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from DFGNN.parallel import OverlappedGradSync
        torch.manual_seed(0)

        class Two(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.blk = torch.nn.ModuleDict({"a": torch.nn.Linear(4, 1, bias=False), "b": torch.nn.Linear(4, 1, bias=False)})
                self.other = torch.nn.Linear(4, 1, bias=False)

        out = {}
        for overlap in (True, False):
            model = Two()
            sync = OverlappedGradSync(model, overlap=overlap)
            x = torch.arange(4.0)[None] + rank
            # step 1: every rank uses blk.a; only rank 0 uses blk.b (a bucket that is only partly filled on rank 1) and
            # only rank 1 uses `other` (a bucket no gradient reaches on rank 0)
            loss = model.blk["a"](x).sum() + (model.blk["b"](x).sum() if rank == 0 else 0) + (model.other(x).sum() if rank == 1 else 0)
            loss.backward()
            sync.finish()
            g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
            # step 2 (grads kept, zero_grad(set_to_none=False)): nobody uses blk.b now -- its slot must travel as zeros
            model.zero_grad(set_to_none=False)
            (2 * model.blk["a"](x).sum()).backward()
            sync.finish()
            g2 = {n: p.grad.clone() for n, p in model.named_parameters()}
            refused = False
            if overlap:
                model.other(x).sum().backward()         # completes the last bucket: its all-reduce is now in flight
                try:
                    model.other(x).sum().backward()     # a second backward before finish()
                except RuntimeError as e:
                    refused = "finish()" in str(e)
                sync.finish()
            out[overlap] = (g1, g2, refused)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_gradient_exchange_with_buckets_only_some_ranks_fill():
    """OverlappedGradSync when a bucket gets gradients for only some of its parameters, or none at all, on some rank:
    every rank still issues the same collectives (no hang), missing slots travel as zeros (not as the previous step's
    values), and a second backward while a bucket's all-reduce is pending is refused."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_partial_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    x0, x1 = torch.arange(4.0)[None], torch.arange(4.0)[None] + 1
    for rank in range(world):
        for overlap in (True, False):
            g1, g2, refused = ret[rank][overlap]
            assert torch.equal(g1["blk.a.weight"], x0 + x1)
            assert torch.equal(g1["blk.b.weight"], x0)            # rank 1 contributed zeros
            assert torch.equal(g1["other.weight"], x1)            # rank 0 had no gradient at all: it receives rank 1's
            assert torch.equal(g2["blk.a.weight"], 2 * (x0 + x1))
            assert torch.equal(g2["blk.b.weight"], torch.zeros(1, 4))   # not step 1's slot again
            assert torch.equal(g2["other.weight"], torch.zeros(1, 4))
            assert refused == overlap
