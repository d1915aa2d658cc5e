#!/usr/bin/env python3
"""bench.py -- headline benchmark: edges/s (fwd+bwd) of the fused GT conv ('hyper'), PATTERN-like
batch bs=1024, dim=128, heads=1 (BASELINE.json configs[2]), on N GPUs of one node.

One "step" = one pass of the hot path over one resident batch: GTConvFuse_hyper forward
(writes out + attn_edge) + its autograd backward (dQ, dK, dV) at the operator boundary, exactly what
DFGNN/script/train/train_batch_graph_timing.py times per layer in the reference.
Inputs (CSR/COO/CSC index arrays, Q, K, V, dO) are resident in HBM before the timed region.

N > 1: one process per GPU (torch.distributed, RCCL).  Whole graphs are the shard unit; each rank
owns its own bs=1024 batch (weak scaling) and fwd+bwd needs no data-path collective (dQ/dK/dV of a
graph depend only on that graph).  The forward-output all-gather that an inference caller may want is
timed separately and reported under "inference_allgather" -- it is never part of `value`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; ~6.3 TB/s is the measured copy rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--heads", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-graphs", type=int, default=128)
    # rehearsal only: "gloo" lets several ranks share ONE GPU (collectives on host tensors) to exercise the N > 1
    # code path on a single-GPU box; the driver's multi-GPU runs use the default (RCCL).
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    return ap.parse_args()


def algorithmic_bytes(m, nnz, h, f):
    """Compulsory HBM bytes per launch (each distinct input read once, each output written once);
    SURVEY.md 8(d), split per kernel (DESIGN.md 'Roofline')."""
    D = h * f
    return {
        # reads Q,K,V, row_ptr, col_ind, rows, val; writes out, attn_edge
        "gt_hyper_fwd": 16 * m * D + 12 * nnz + 4 * (m + 1) + 4 * h * nnz,
        # whole backward: reads Q,K,V,dO, attn_edge, CSR + CSC index arrays; writes dQ,dK,dV; the
        # reference's grad_edge round trip (8 h nnz) is part of the figure even though the resident kernel avoids it
        "gt_bwd": 28 * m * D + 12 * h * nnz + 16 * nnz + 8 * (m + 1),
        # the two general (plan-less) backward launches, each reading its inputs once
        "gt_bwd_rows(general)": 16 * m * D + 8 * h * nnz + 12 * nnz + 4 * (m + 1),
        "gt_bwd_cols(general)": 16 * m * D + 8 * h * nnz + 12 * nnz + 4 * (m + 1),
    }


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the fused kernels have no CPU fallback)"
    if args.dist_backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")  # where collective buffers live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    import dfgnn_native
    import fused_gtconv
    from DFGNN.layers import preprocess_Hyper_fw_bw
    from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
    from DFGNN.utils import synthetic as S

    h, f = args.heads, args.dim // args.heads
    # ---- workload: this rank's shard = its own PATTERN-like batch (whole graphs), resident in HBM
    g_host = S.pattern_like(batch_size=args.batch_size, seed=1 + rank)
    g = g_host.to(dev)
    A, rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem = preprocess_Hyper_fw_bw(g)
    m, nnz = g.num_nodes(), g.num_edges()
    Q, K, V = (t.requires_grad_(True) for t in S.gt_features(m, h, f, seed=100 + rank, device=dev))
    dO = torch.randn(m, h, f, generator=torch.Generator().manual_seed(7 + rank)).to(dev)

    def step():
        out = GTConvFuse_hyper(rows, row_ptr, col_ind, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
        return torch.autograd.grad(out, (Q, K, V), dO)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        e = torch.tensor([nnz], dtype=torch.int64, device=cdev)
        dist.all_reduce(e)
        total_edges = int(e.item())
    else:
        total_edges = nnz
    ms_per_step = elapsed / args.steps * 1e3
    value = total_edges * args.steps / elapsed

    # ---- per-kernel attribution with device events on the launch stream (same inputs, K launches each)
    L = dfgnn_native.lib()
    P = lambda t: t.data_ptr()  # noqa: E731
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.no_grad():
        out = torch.empty_like(Q)
        attn = torch.empty(h, nnz, device=dev)
        gedge = torch.empty(h, nnz, device=dev)
        dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    from _binding_util import get_plan, val_ptr
    plan, plan_meta, _ = get_plan(row_ptr, col_ind, f, fused_gtconv.USE_BLOCK_PLAN)
    VP = val_ptr(val)  # NULL for unit edge values, exactly as the binding passes it (fused_gtconv.py)
    calls = {
        "gt_hyper_fwd": lambda: L.dfgnn_gt_hyper_fwd(m, nnz, h, f, P(row_ptr), P(col_ind), P(rows), VP, P(Q), P(K),
                                                     P(V), P(attn), None, P(out), plan, plan_meta, stream),
        "gt_bwd": lambda: L.dfgnn_gt_bwd(m, nnz, h, f, P(row_ptr), P(col_ind), P(rows), VP, P(col_ptr), P(row_ind),
                                         P(val_idx), P(Q), P(K), P(V), P(attn), P(dO), P(gedge), P(dQ), P(dK), P(dV),
                                         plan, plan_meta, stream),
        "gt_bwd_rows(general)": lambda: L.dfgnn_gt_bwd_rows(m, nnz, h, f, P(row_ptr), P(col_ind), P(rows), P(val), P(K), P(V),
                                                   P(attn), P(dO), P(gedge), P(dQ), stream),
        "gt_bwd_cols(general)": lambda: L.dfgnn_gt_bwd_cols(m, nnz, h, f, P(val), P(col_ptr), P(row_ind), P(val_idx), P(Q),
                                                   P(attn), P(gedge), P(dO), P(dK), P(dV), stream),
    }
    kernel_us = {}
    reps = max(10, args.steps)
    for name, call in calls.items():
        for _ in range(3):
            assert call() == 0
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record()
            assert call() == 0
            b.record()
        torch.cuda.synchronize()
        kernel_us[name] = float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3
    abytes = algorithmic_bytes(m, nnz, h, f)
    dom = max(("gt_hyper_fwd", "gt_bwd"), key=kernel_us.get)  # the launches the timed step actually runs
    achieved = abytes[dom] / (kernel_us[dom] * 1e-6) / 1e9
    # HBM bytes per launch from PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes on this
    # exact workload; profiles/r01_pmc_dense_kernels.json).  bench.py cannot collect counters itself, so the figure
    # is reported only when the workload matches the profiled one, else null.
    traffic = None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_dense_kernels.json")))
        if f"m={m}, nnz={nnz}" in prof["workload"] and h == 1 and f == 128:
            key = {"gt_hyper_fwd": "gt_dense_fwd_kernel", "gt_bwd": "gt_dense_bwd_kernel"}[dom]
            traffic = int(prof["traffic"][key]["total_bytes"])
    except Exception:
        traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes": abytes[dom], "avg_us": round(kernel_us[dom], 2),
                "all_kernels": {k: {"avg_us": round(v, 2), "algorithmic_bytes": abytes[k],
                                    "achieved_GBs": round(abytes[k] / (v * 1e-6) / 1e9, 1)}
                                for k, v in kernel_us.items()}}

    # ---- secondary figures on the same resident batch (never part of `value`): the GAT training pair (SURVEY.md 8f
    #      rank 1), the per-batch preprocessing the reference counts inside an epoch (8f rank 2), and the timed step
    #      replayed as one HIP graph
    secondary = None
    if rank == 0 and world == 1:
        import dfgnn_preprocess
        import fused_gatconv
        from _binding_util import build_plan
        from DFGNN.utils import GraphedStep

        def ev_us(fn, reps=10):
            for _ in range(3):
                fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3

        def wall_ms(fn, reps=10):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        with torch.no_grad():
            ar, ac, X = S.gat_features(m, h, f, seed=6, device=dev)
            go, emax, esum, mask = fused_gatconv.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0)
            gat_f = ev_us(lambda: fused_gatconv.gat_forward(ar, ac, row_ptr, col_ind, 0.2, X, 0.0))
            gat_b = ev_us(lambda: fused_gatconv.gat_backward(0.2, 0.0, row_ptr, col_ind, col_ptr, row_ind, val_idx, emax,
                                                             esum, mask, X, ar, ac, dO))
        src, dst = g.edges()

        def raw_step():  # the launches of the timed step as explicit operator calls (no autograd bookkeeping)
            with torch.no_grad():
                o, at = fused_gtconv.gt_hyper_forward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V)
                return fused_gtconv.gt_backward(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, at, dO)

        graphed = GraphedStep(raw_step)
        secondary = {
            "gat_train": {"fwd_us": round(gat_f, 1), "bwd_us": round(gat_b, 1),
                          "edges_per_s": nnz / ((gat_f + gat_b) * 1e-6)},
            "preprocess_per_batch_ms": {
                "coo_to_csr_csc_native": round(wall_ms(lambda: dfgnn_preprocess.coo_to_hyper(src, dst, m, csc=True)), 3),
                "block_plan": round(wall_ms(lambda: build_plan(row_ptr, col_ind, f)), 3)},
            "step_as_hipgraph_ms": round(wall_ms(graphed.replay, reps=max(10, args.steps)), 4),
        }

    # ---- forward-output all-gather (the inference-side exchange step), timed separately
    gather = None
    if world > 1:
        mx = torch.tensor([m], dtype=torch.int64, device=cdev)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        m_pad = int(mx.item())
        send = torch.zeros(m_pad, h, f, device=cdev)
        recv = torch.empty(world * m_pad, h, f, device=cdev)

        def infer_step():
            o = fused_gtconv.gt_hyper_inference(row_ptr, col_ind, rows, val, smem, Q.detach(), K.detach(), V.detach())[0]
            send[:m].copy_(o)
            if args.dist_backend == "nccl":
                dist.all_gather_into_tensor(recv, send)
            else:
                dist.all_gather(list(recv.chunk(world)), send)

        for _ in range(3):
            infer_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            infer_step()
        barrier()
        tg = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        gather = {"ms_per_step": round(float(tg.item()) / args.steps * 1e3, 4),
                  "edges_per_s": total_edges * args.steps / float(tg.item()),
                  "gathered_MB_per_rank": round(world * m_pad * h * f * 4 / 1e6, 1)}

    # ---- CPU baseline: the oracle's fp32 OpenMP build on a bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (the contract: rank 0 at N = 1 only)
        import oracle
        oracle.build()
        ng = min(args.cpu_sample_graphs, args.batch_size)
        sizes = g_host.batch_num_nodes().numpy()
        n_s = int(sizes[:ng].sum())
        rp = row_ptr[: n_s + 1].cpu().numpy()
        e_s = int(rp[-1])
        ci = col_ind[:e_s].cpu().numpy()
        vl = val[:e_s].cpu().numpy()
        q, k, v, do = (t[:n_s].detach().cpu().numpy() for t in (Q, K, V, dO))
        oracle.gt_forward(rp, ci, vl, q, k, v, acc="f32")  # warm-up
        t0 = time.perf_counter()
        n_rep = 0
        while True:
            oracle.gt_forward(rp, ci, vl, q, k, v, want_attn=True, acc="f32")
            oracle.gt_backward(rp, ci, vl, q, k, v, do, acc="f32")
            n_rep += 1
            if time.perf_counter() - t0 > 10.0 or n_rep >= 200:
                break
        dt = time.perf_counter() - t0
        cpu = {"value": e_s * n_rep / dt, "unit": "edges/s", "cores": oracle.num_threads(), "kind": "port",
               "sample": f"first {ng} graphs of the rank-0 batch ({n_s} nodes, {e_s} edges), fwd+bwd, "
                         f"{n_rep} reps in {dt:.1f}s, fp32 C/OpenMP oracle"}

    if rank == 0:
        line = {
            "metric": "edges/s (fwd+bwd) fused GT conv, PATTERN bs=1024 dim=128",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"GT conv 'hyper' fwd+bwd, PATTERN-like batch bs={args.batch_size} "
                                   f"dim={args.dim} heads={h} (BASELINE.json configs[2])",
                       "nodes_per_gpu": m, "edges_per_gpu": nnz, "total_edges": total_edges,
                       "parallelism": f"graph-sharded x{world}, no data-path collective"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if gather:
            line["inference_allgather"] = gather
        if secondary:
            line["secondary"] = secondary
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
