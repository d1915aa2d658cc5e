#!/usr/bin/env python3
"""The operator in its real calling pattern (reference: an 8-layer stack trained with Adam,
DFGNN/script/train/train_batch_graph_timing.py:32-53, 146-196): fresh PATTERN-like batches every step, preprocessing
(COO -> CSR / CSC + block plan) paid per batch, L SparseMHA_forward layers sharing the batch structure, MSE loss,
Adam.  Prints one JSON line per mode (fused operators vs the non-fused torch branch) with the per-step time split the
reference reports (preprocess / forward / backward+update) and the loss after the last step.
usage: python3 tools/train_stack.py [--layers 8] [--batch-size 256] [--dim 128] [--steps 12] [--batches 4]

Data-parallel form (one process per GPU, `python -m torch.distributed.run --nproc-per-node N tools/train_stack.py
--modes fused`): every rank trains on its shard of each batch (whole graphs, DFGNN/parallel/sharding.py), the fused
convolution exchanges nothing, and the weight gradients are all-reduced bucket by bucket while the backward is still
running (DFGNN/parallel/overlap.py); the line then carries the step time with the exchange overlapped and with it issued
after the backward.  --dist-backend gloo rehearses it with several ranks on one GPU."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch import nn  # noqa: E402

from DFGNN.layers import SparseMHA_forward, preprocess_Hyper_fw_bw  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402


class Stack(nn.Module):
    def __init__(self, dim, layers):
        super().__init__()
        self.inproj = nn.Linear(dim, dim)
        self.layers = nn.ModuleList(SparseMHA_forward(dim, dim, 1) for _ in range(layers))
        self.out = nn.Linear(dim, 1)

    def forward(self, params, x, fuse):
        h = self.inproj(x)
        for layer in self.layers:
            h = h + layer(params, h, fuse)          # residual, as the reference's GTModel does around its MHA blocks
        return self.out(h)


def run(args, fuse, dev, overlap=None):
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    torch.manual_seed(0)
    model = Stack(args.dim, args.layers).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sync = None
    if world > 1:
        from DFGNN.parallel import OverlappedGradSync, shard_graph
        sync = OverlappedGradSync(model, overlap=bool(overlap))
    graphs = [S.pattern_like(batch_size=args.batch_size, seed=10 + b) for b in range(args.batches)]
    global_nodes = [g.num_nodes() for g in graphs]  # squared errors / nodes of the WHOLE batch: shard gradients sum to its mean-loss gradient
    if world > 1:
        graphs = [shard_graph(g, rank, world)[0] for g in graphs]
    graphs = [g.to(dev) for g in graphs]
    feats = [torch.randn(g.num_nodes(), args.dim, device=dev, generator=torch.Generator(dev).manual_seed(b))
             for b, g in enumerate(graphs)]
    target = [torch.randn(g.num_nodes(), 1, device=dev, generator=torch.Generator(dev).manual_seed(100 + b))
              for b, g in enumerate(graphs)]
    t_prep = t_fwd = t_bwd = 0.0
    loss_v, edges = None, 0
    for step in range(args.warmup + args.steps):
        b = step % args.batches
        g = graphs[b]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        params = preprocess_Hyper_fw_bw(g, fused=True)      # fresh tensors: the block plan is rebuilt as well
        if not fuse:
            params = (params[0],) + (None,) * 8
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss = nn.functional.mse_loss(model(params, feats[b], fuse), target[b], reduction="sum") / global_nodes[b]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if sync is not None:
            sync.finish()
        opt.step()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        if step >= args.warmup:
            t_prep += t1 - t0
            t_fwd += t2 - t1
            t_bwd += t3 - t2
            edges += g.num_edges() * args.layers
        loss_v = float(loss.detach())
    n = args.steps
    return {"mode": ("fused" if fuse else "torch baseline (fuse=False)") +
            ("" if world == 1 else f", {world} ranks, weight-gradient all-reduce " + ("overlapped with the backward" if overlap else "after the backward")),
            "layers": args.layers, "batch_size": args.batch_size,
            "dim": args.dim, "steps": n, "preprocess_ms": t_prep / n * 1e3, "forward_ms": t_fwd / n * 1e3,
            "backward_update_ms": t_bwd / n * 1e3, "step_ms": (t_prep + t_fwd + t_bwd) / n * 1e3,
            "layer_edges_per_s": edges / (t_prep + t_fwd + t_bwd), "final_loss": loss_v,
            # after training: L1 norm of the attention projections' total update direction (sensitive to the operator's
            # gradients, unlike the loss, which the random targets dominate)
            "qkv_weight_l1": float(sum(p.detach().abs().sum() for n_, p in model.named_parameters() if "_proj" in n_)),
            "qkv_weights": torch.cat([p.detach().flatten() for n_, p in model.named_parameters() if "q_proj.weight" in n_])
            if args.layers <= 3 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--modes", default="fused,baseline")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.dist_backend, **({"device_id": dev} if args.dist_backend == "nccl" else {}))
        for overlap in (True, False):
            r = run(args, True, dev, overlap)
            r.pop("qkv_weights")
            if dist.get_rank() == 0:
                print(json.dumps(r), flush=True)
        dist.destroy_process_group()
        return
    res = []
    for mode in args.modes.split(","):
        res.append(run(args, mode == "fused", dev))
        res[-1].pop("qkv_weights")
        print(json.dumps(res[-1]), flush=True)
    if len(res) == 2:
        print(json.dumps({"speedup_step": res[1]["step_ms"] / res[0]["step_ms"],
                          "loss_abs_diff": abs(res[0]["final_loss"] - res[1]["final_loss"]),
                          "qkv_weight_l1_rel_diff": abs(res[0]["qkv_weight_l1"] - res[1]["qkv_weight_l1"])
                          / res[1]["qkv_weight_l1"]}), flush=True)


if __name__ == "__main__":
    main()
