#!/usr/bin/env python3
"""bench.py -- headline benchmark: edges/s (fwd+bwd) of the fused GT conv ('hyper'), PATTERN-like
batch bs=1024, dim=128, heads=1 (BASELINE.json configs[2]), on N GPUs of one node.

One "step" = one pass of the hot path over one resident batch: GTConvFuse_hyper forward + its autograd backward
(dQ, dK, dV) at the operator boundary, exactly what DFGNN/script/train/train_batch_graph_timing.py times per layer in
the reference.  FusedGTFunction_hyper has two forms of the pair on a batch like this one (every
member graph a dense range of the block plan, unit edge values): the reference's -- the forward writes attn_edge, the
backward reads it -- and one that saves two floats per (row, head) and recomputes the attention in the backward
(include/dfgnn.h: dfgnn_gt_hyper_fwd_stats / dfgnn_gt_bwd_stats).  It takes the faster one: the attn_edge pair at one
head (this headline) -- with the values in RANK order (by column within a row: dfgnn_gt_hyper_fwd_ranked /
dfgnn_gt_bwd_ranked; the order is internal to the autograd function, and this one lets the forward find an edge's slot
from the plan's bitmap instead of the edge list) --, the statistics pair from two heads on (--heads 2/4/8).  All are
timed; what the step does not launch is reported under `secondary` (`row_statistics_pair` / `attn_edge_pair`).
Inputs (CSR/COO/CSC index arrays, Q, K, V, dO) are resident in HBM before the timed region.

N > 1: one process per GPU (torch.distributed, RCCL).  Launched by the driver as `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE from the environment); run plainly as
`python bench.py --gpus N` it starts its own N rank processes first -- before anything in this process touches a GPU --
and rank 0's line is the output.  Whole graphs are the shard unit and fwd+bwd needs no
data-path collective (dQ/dK/dV of a graph depend only on that graph).
  --scaling weak   (default, the driver contract): every rank owns its own bs=1024 batch; `value` = all edges / time.
  --scaling strong: the ONE seed-1 bs=1024 batch is cut into N shards of whole graphs, balanced by edge
                   count (DFGNN/parallel/sharding.py:shard_graph); `value` = its edges / max-over-ranks time.
With N > 1 the line also carries the other mode's measurement ("strong_scaling" / "weak_scaling") and the
forward-output all-gather an inference caller may want ("inference_allgather": kernel + RCCL all-gather of the padded
per-shard outputs) -- never part of `value`.

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
PMC_PROFILE = os.path.join("profiles", "r03_pmc_dense_kernels.json")  # HBM bytes per launch from rocprofv3 --pmc passes
SETUP_STEPS = 40  # untimed steps inside Workload(): plan build, allocator pools, device clocks (~10 ms of continuous work; reported as "setup_steps")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=1024)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--heads", type=int, default=1)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c4", action="store_true", help="skip the reddit-like GAT 'tiling' secondary figure")
    ap.add_argument("--cpu-sample-graphs", type=int, default=128)
    # rehearsal only: "gloo" lets several ranks share ONE GPU (collectives on host tensors) to exercise the N > 1
    # code path on a single-GPU box; the driver's multi-GPU runs use the default (RCCL).
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    # no GPU work at all: ranks are started / joined, the process group comes up (gloo), the collectives the timed region
    # uses are exercised on dummy numbers, rank 0 prints a line with "dry_run": true and no value -- what the CPU suite runs
    ap.add_argument("--dry-run", action="store_true")
    return ap.parse_args()


def algorithmic_bytes_stats(m, nnz, h, f):
    """Compulsory HBM bytes per launch of the statistics-saving pair (each distinct input read once, each output written
    once): what SURVEY.md 8(d) counts MINUS everything this pair no longer moves -- attn_edge (4 h nnz written, 4 h nnz
    read), grad_edge (8 h nnz), rows / col_ind / val / the CSC arrays -- PLUS what it moves instead: the row statistics
    (2 floats per (row, head), written once and read once) and the plan's edge bitmaps (32 bytes per node: the out-edges;
    the multi-head backward also reads the in-edges)."""
    D = h * f
    return {
        "gt_hyper_fwd_stats": 16 * m * D + 8 * m * h + 32 * m,
        "gt_bwd_stats": 28 * m * D + 8 * m * h + 32 * m * (2 if h > 1 else 1),
    }


def algorithmic_bytes(m, nnz, h, f):
    """Compulsory HBM bytes per launch of the attn_edge pair (each distinct input read once, each output written once);
    SURVEY.md 8(d), split per kernel (DESIGN.md 'Roofline')."""
    D = h * f
    return {
        # reads Q,K,V, row_ptr, col_ind, rows, val; writes out, attn_edge
        "gt_hyper_fwd": 16 * m * D + 12 * nnz + 4 * (m + 1) + 4 * h * nnz,
        # whole backward: reads Q,K,V,dO, attn_edge, CSR + CSC index arrays; writes dQ,dK,dV; the
        # reference's grad_edge round trip (8 h nnz) is part of the figure even though the resident kernel avoids it
        "gt_bwd": 28 * m * D + 12 * h * nnz + 16 * nnz + 8 * (m + 1),
    }


def algorithmic_bytes_ranked(m, nnz, h, f):
    """The attn_edge pair with the values in rank order (dfgnn_gt_hyper_fwd_ranked / dfgnn_gt_bwd_ranked).  Forward: what it
    moves -- Q, K, V in, out and attn_edge out, row_ptr and the out-edge bitmap in (no rows / col_ind / val: 12 nnz less than
    SURVEY.md 8(d) counts).  Backward: the kernel and the figure of the attn_edge pair."""
    D = h * f
    return {
        "gt_hyper_fwd_ranked": 16 * m * D + 4 * h * nnz + 4 * (m + 1) + 32 * m,
        "gt_bwd_ranked": algorithmic_bytes(m, nnz, h, f)["gt_bwd"],
    }


def ev_us(fn, reps=10, warm=3, steady_ms=8.0):
    """Mean device time of fn() in microseconds (events on torch's current stream = the launch stream).  After the `warm`
    calls fn() is repeated until about `steady_ms` of device work lie behind the measurement: a device that idled through the
    host work before this call runs its first milliseconds at ramping clocks (the same effect as in timed(), main())."""
    a0, b0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record()
    for _ in range(warm):
        fn()
    b0.record()
    torch.cuda.synchronize()
    per_call_ms = max(a0.elapsed_time(b0) / max(warm, 1), 1e-3)
    for _ in range(min(400, int(steady_ms / per_call_ms))):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3


def wall_ms(fn, reps=10, warm=3, steady_ms=8.0):
    """Wall time per fn() call in ms; like ev_us, about `steady_ms` of device work lie right behind the timed calls."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    per_call_ms = max((time.perf_counter() - t0) * 1e3 / max(warm, 1), 1e-3)
    for _ in range(min(400, int(steady_ms / per_call_ms))):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


class Workload:
    """One rank's resident batch: graph structure (hyper format + CSC), features and the step closure."""

    def __init__(self, g_host, h, f, feat_seed, dev):
        from DFGNN.layers import preprocess_Hyper_fw_bw
        from DFGNN.operators.fused_gtconv import GTConvFuse_hyper
        from DFGNN.utils import synthetic as S
        self.g_host, self.g = g_host, g_host.to(dev)
        (_, self.rows, self.row_ptr, self.col_ind, self.val, self.col_ptr, self.row_ind, self.val_idx,
         self.smem) = preprocess_Hyper_fw_bw(self.g)
        self.m, self.nnz = self.g.num_nodes(), self.g.num_edges()
        self.Q, self.K, self.V = (t.requires_grad_(True) for t in S.gt_features(self.m, h, f, seed=feat_seed, device=dev))
        self.dO = torch.randn(self.m, h, f, generator=torch.Generator().manual_seed(7 + feat_seed)).to(dev)
        self._op = GTConvFuse_hyper
        # part of setting the workload up, like the preprocessing above: the first call builds the block plan, the next few
        # fill the allocator's pools and bring host and device clocks up (the first ~30 operator calls of a process run
        # slower, see tests/tools/bench_configs.py)
        import gc
        gc.collect()   # here, with the device idle anyway, not between warm-up and timing (see timed())
        for _ in range(SETUP_STEPS):
            self.step()
        torch.cuda.synchronize(dev)

    def step(self):
        out = self._op(self.rows, self.row_ptr, self.col_ind, self.val, self.col_ptr, self.row_ind, self.val_idx,
                       self.smem, self.Q, self.K, self.V)
        return torch.autograd.grad(out, (self.Q, self.K, self.V), self.dO)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, a free local port) and return the largest exit code.  Nothing in THIS
    process has touched a GPU yet (importing torch does not), and nothing does: it only waits.  Rank 0 prints the line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


def dry_run(args, world, rank):
    """--dry-run: the launch / rendezvous / collective plumbing of an N-rank run without a GPU."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(e)
        dist.barrier()
        assert int(t.item()) == world and int(e.item()) == world * (world + 1) // 2
    if rank == 0:
        print(json.dumps({"metric": "edges/s (fwd+bwd) fused GT conv, PATTERN bs=1024 dim=128", "value": None, "unit": "edges/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "scaling": args.scaling if world > 1 else "weak"}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a line "
                  f"for a job of another size", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, world, rank)
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
    from DFGNN.parallel import shard_graph
    from DFGNN.utils import synthetic as S

    h, f = args.heads, args.dim // args.heads

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def allsum(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.int64, device=cdev)
        dist.all_reduce(t)
        return int(t.item())

    def make(scaling):
        """This rank's workload: its own batch (weak) or its shard of the seed-1 batch (strong)."""
        if scaling == "weak" or world == 1:
            return Workload(S.pattern_like(batch_size=args.batch_size, seed=1 + (rank if scaling == "weak" else 0)), h, f,
                            100 + rank, dev)
        full = S.pattern_like(batch_size=args.batch_size, seed=1)
        sub, _ = shard_graph(full, rank, world)
        return Workload(sub, h, f, 100 + rank, dev)

    def timed(w, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` timed ones between barriers (barrier = device synchronise + process
        barrier).  The collector is held off for the timed steps: a collection pass in a 13 ms region shows up as +10 %."""
        import gc
        mode = os.environ.get("DFGNN_BENCH_GC", "setup")   # (A/B switch: where the collection pass goes)
        if mode == "early":
            gc.collect()
        if mode != "late":
            # the collection pass is NOT between warm-up and timing: it takes tens of milliseconds of host time, and with the
            # device idle that long the first timed steps run at ramping clocks (seen as +2..4 % on a 20-step region).  It runs
            # when the workload is set up, ahead of its set-up steps (Workload.__init__)
            gc.disable()
        for _ in range(warmup):
            w.step()
        barrier()
        if mode == "late":
            gc.collect()
            gc.disable()
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                w.step()
            barrier()
            return allmax(time.perf_counter() - t0)
        finally:
            gc.enable()

    # ---- the timed region -------------------------------------------------------------------------------------------
    W = make(args.scaling)
    m, nnz = W.m, W.nnz
    elapsed = timed(W, args.steps, args.warmup)
    total_edges = allsum(nnz)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_edges * args.steps / elapsed

    # ---- the other scaling mode and the inference-side exchange step (N > 1 only) -------------------------------
    other = gather = None
    if world > 1:
        try:   # (secondary figures: a failure here must not cost the line its headline; every rank takes the same path)
            other_mode = "strong" if args.scaling == "weak" else "weak"
            Wo = make(other_mode)
            t_o = timed(Wo, args.steps, args.warmup)
            e_o = allsum(Wo.nnz)
            other = {"ms_per_step": round(t_o / args.steps * 1e3, 4), "edges_per_s": e_o * args.steps / t_o,
                     "total_edges": e_o, "edges_this_rank": Wo.nnz,
                     "note": ("the seed-1 bs=%d batch cut into %d shards of whole graphs (shard_graph, edge-balanced); kernel time "
                              "only, no collective on the fwd+bwd path" % (args.batch_size, world)) if other_mode == "strong"
                     else "every rank its own bs=%d batch" % args.batch_size}
            Ws = W if args.scaling == "strong" else Wo      # the all-gather belongs to the sharded-batch picture
            mx = int(allmax(float(Ws.m)))
            send = torch.zeros(mx, h, f, device=cdev)
            recv = torch.empty(world * mx, h, f, device=cdev)

            def infer_step():
                o = fused_gtconv.gt_hyper_inference(Ws.row_ptr, Ws.col_ind, Ws.rows, Ws.val, Ws.smem, Ws.Q.detach(),
                                                    Ws.K.detach(), Ws.V.detach())[0]
                send[:Ws.m].copy_(o)
                if args.dist_backend == "nccl":
                    dist.all_gather_into_tensor(recv, send)
                else:
                    dist.all_gather(list(recv.chunk(world)), send)

            def infer_only():
                fused_gtconv.gt_hyper_inference(Ws.row_ptr, Ws.col_ind, Ws.rows, Ws.val, Ws.smem, Ws.Q.detach(), Ws.K.detach(),
                                                Ws.V.detach())

            times = {}
            for name, fn in (("kernel_plus_allgather", infer_step), ("kernel_only", infer_only)):
                for _ in range(3):
                    fn()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fn()
                barrier()
                times[name] = allmax(time.perf_counter() - t0) / args.steps * 1e3
            e_s = allsum(Ws.nnz)
            gather = {"ms_per_step": round(times["kernel_plus_allgather"], 4), "ms_kernel_only": round(times["kernel_only"], 4),
                      "edges_per_s": e_s / (times["kernel_plus_allgather"] * 1e-3), "gathered_MB_per_rank": round(world * mx * h * f * 4 / 1e6, 1),
                      "workload": "forward (inference) of the sharded seed-1 batch, outputs padded to the largest shard, one "
                                  "all_gather_into_tensor"}

        except Exception as exc:  # noqa: BLE001
            other = other or {"error": repr(exc)}
            gather = {"error": repr(exc)}

    # ---- per-kernel attribution with device events on the launch stream (same inputs) -----------------------------
    L = dfgnn_native.lib()
    P = lambda t: t.data_ptr()  # noqa: E731
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.no_grad():
        out = torch.empty_like(W.Q)
        attn = torch.empty(h, nnz, device=dev)
        gedge = torch.empty(h, nnz, device=dev)
        dQ, dK, dV = torch.empty_like(W.Q), torch.empty_like(W.Q), torch.empty_like(W.Q)
    from _binding_util import get_plan, val_ptr
    plan, plan_meta, _ = get_plan(W.row_ptr, W.col_ind, f, fused_gtconv.USE_BLOCK_PLAN)
    VP = val_ptr(W.val)  # NULL for unit edge values, exactly as the binding passes it (fused_gtconv.py)

    uses_stats = fused_gtconv.gt_stats_pair_chosen(W.row_ptr, W.col_ind, W.val, W.Q)  # what the timed step launches
    stats_can = fused_gtconv.gt_stats_pair_applies(W.row_ptr, W.col_ind, W.val, W.Q)  # ... and whether the other pair could run
    with torch.no_grad():
        rmax, rsum = torch.empty(m, h, device=dev), torch.empty(m, h, device=dev)

    def fwd_stats_call():
        return L.dfgnn_gt_hyper_fwd_stats(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), None, P(W.Q), P(W.K), P(W.V), P(rmax), P(rsum),
                                          P(out), plan, plan_meta, stream)

    def bwd_stats_call():
        return L.dfgnn_gt_bwd_stats(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), None, P(W.Q), P(W.K), P(W.V), P(rmax), P(rsum),
                                    P(W.dO), P(dQ), P(dK), P(dV), plan, plan_meta, stream)

    def fwd_call(vp):
        return lambda: L.dfgnn_gt_hyper_fwd(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(W.rows), vp, P(W.Q), P(W.K), P(W.V),
                                            P(attn), None, P(out), plan, plan_meta, stream)

    def bwd_call(vp):
        return lambda: L.dfgnn_gt_bwd(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(W.rows), vp, P(W.col_ptr), P(W.row_ind),
                                      P(W.val_idx), P(W.Q), P(W.K), P(W.V), P(attn), P(W.dO), P(gedge), P(dQ), P(dK), P(dV),
                                      plan, plan_meta, stream)

    def checked(call):
        def run():
            assert call() == 0
        return run

    reps = max(10, args.steps)
    attn_us = {"gt_hyper_fwd": ev_us(checked(fwd_call(VP)), reps), "gt_bwd": ev_us(checked(bwd_call(VP)), reps)}
    attn_bytes = algorithmic_bytes(m, nnz, h, f)
    # one head, all dense, unit values: the attn_edge pair in rank order is what the step launches
    ranked_can = fused_gtconv.gt_ranked_pair_chosen(W.row_ptr, W.col_ind, W.val, W.Q)
    ranked_us = None
    if ranked_can is not None:
        ranked_us = {
            "gt_hyper_fwd_ranked": ev_us(checked(lambda: L.dfgnn_gt_hyper_fwd_ranked(
                m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(W.Q), P(W.K), P(W.V), P(attn), P(out), plan, plan_meta, stream)), reps),
            "gt_bwd_ranked": ev_us(checked(lambda: L.dfgnn_gt_bwd_ranked(
                m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(W.Q), P(W.K), P(W.V), P(attn), P(W.dO), P(dQ), P(dK), P(dV), plan,
                plan_meta, stream)), reps)}
    stats_us, stats_bytes = None, algorithmic_bytes_stats(m, nnz, h, f)
    if stats_can is not None:
        stats_us = {"gt_hyper_fwd_stats": ev_us(checked(fwd_stats_call), reps), "gt_bwd_stats": ev_us(checked(bwd_stats_call), reps)}
    if uses_stats is not None:
        kernel_us, abytes = stats_us, stats_bytes
    elif ranked_us is not None:
        kernel_us, abytes = ranked_us, algorithmic_bytes_ranked(m, nnz, h, f)
    else:
        kernel_us, abytes = attn_us, attn_bytes
    # edge values other than ones: the statistics pair with the values in the plan's dense form (what FusedGTFunction_hyper
    # launches for a weighted batch: csrc/gt_dense_stats_w.hip); the dense form is built once per (plan, val)
    weighted = None
    if stats_can is not None and rank == 0:
        from _binding_util import plan_dense_weights
        with torch.no_grad():
            wval = torch.rand(nnz, 1, device=dev) + 0.5
            t0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0[0].record()
            wd = plan_dense_weights(stats_can, W.row_ptr, wval)
            t0[1].record()
            torch.cuda.synchronize()
        wf = ev_us(checked(lambda: L.dfgnn_gt_hyper_fwd_stats(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(wd), P(W.Q), P(W.K), P(W.V),
                                                              P(rmax), P(rsum), P(out), plan, plan_meta, stream)), reps)
        wb = ev_us(checked(lambda: L.dfgnn_gt_bwd_stats(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(wd), P(W.Q), P(W.K), P(W.V),
                                                        P(rmax), P(rsum), P(W.dO), P(dQ), P(dK), P(dV), plan, plan_meta, stream)), reps)
        weighted = {"what": "fwd + bwd with edge values (uniform in [0.5, 1.5)) on the matrix cores: statistics pair, values in the "
                            "plan's dense form (256 floats per node, built once per (plan, val))",
                    "fwd_us": round(wf, 2), "bwd_us": round(wb, 2), "edges_per_s": nnz / ((wf + wb) * 1e-6),
                    "dense_form_build_us": round(t0[0].elapsed_time(t0[1]) * 1e3, 1)}
        del wd, wval
    # the same two launches with the edge values passed explicitly (val != NULL): the matrix-core kernels step aside and
    # every product is an fp32 FMA on the VALU (the LDS-resident edge-walking kernels) -- the plain-f32 reference point
    valu_us = {"gt_hyper_fwd": ev_us(checked(fwd_call(P(W.val))), reps), "gt_bwd": ev_us(checked(bwd_call(P(W.val))), reps)}
    dom = max(kernel_us, key=kernel_us.get)  # the launches the timed step actually runs
    achieved = abytes[dom] / (kernel_us[dom] * 1e-6) / 1e9
    # HBM bytes per launch from PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes over this exact
    # workload and build, committed under profiles/): bench.py cannot collect counters itself, so the figure is quoted
    # from that file -- with its name -- only when workload and library build match, else null.
    traffic = traffic_src = None
    try:
        prof = json.load(open(os.path.join(ROOT, PMC_PROFILE)))
        import dfgnn_native
        if (prof.get("m") == m and prof.get("nnz") == nnz and prof.get("h") == h and prof.get("f") == f and
                prof.get("build_id") == dfgnn_native.build_id()):
            key = {"gt_hyper_fwd": "gt_dense_fwd_kernel", "gt_bwd": "gt_dense_bwd_kernel",
                   "gt_hyper_fwd_ranked": "gt_dense_fwd_ranked_kernel", "gt_bwd_ranked": "gt_dense_bwd_kernel",
                   "gt_hyper_fwd_stats": "gt_dense_fwd_stats_kernel", "gt_bwd_stats": "gt_dense_bwd_stats_kernel"}[dom]
            traffic = int(prof["traffic"][key]["total_bytes"])
            traffic_src = PMC_PROFILE + (" (library build %s)" % prof.get("build_id", "?"))
    except Exception:
        traffic = traffic_src = None
    # the box's own device-to-device copy rate next to the datasheet peak (read + written bytes of a 1 GiB copy)
    with torch.no_grad():
        src_buf = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        dst_buf = torch.empty_like(src_buf)
        copy_us = ev_us(lambda: dst_buf.copy_(src_buf), reps=5, warm=2)
        copy_gbs = 2 * src_buf.numel() * 4 / (copy_us * 1e-6) / 1e9
        del src_buf, dst_buf
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "measured_copy_GBs": round(copy_gbs, 1),
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes": abytes[dom], "avg_us": round(kernel_us[dom], 2),
                "all_kernels": {k: {"avg_us": round(v, 2), "algorithmic_bytes": abytes[k],
                                    "achieved_GBs": round(abytes[k] / (v * 1e-6) / 1e9, 1),
                                    "frac": round(abytes[k] / (v * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
                                for k, v in kernel_us.items()}}
    valu_step_us = valu_us["gt_hyper_fwd"] + valu_us["gt_bwd"]
    arithmetic = {
        "io": "f32", "accumulate": "f32",
        "products": "matrix cores (v_mfma_f32_16x16x32_f16): every operand as fp16 hi + lo halves under a power-of-two scale, "
                    "hi*hi + hi*lo + lo*hi accumulated in fp32 (~3 x 2^-24 relative error per product, that of an fp32 FMA "
                    "chain; tests/test_gpu_parity.py::test_dense_kernels_are_fp32_equivalent)",
        "f32_valu_step": {"what": "the attn_edge pair's fwd+bwd launches with every product as an fp32 FMA on the VALU (edge "
                                  "values passed explicitly to dfgnn_gt_hyper_fwd / dfgnn_gt_bwd, which have no matrix-core "
                                  "form for them)",
                          "fwd_us": round(valu_us["gt_hyper_fwd"], 2), "bwd_us": round(valu_us["gt_bwd"], 2),
                          "edges_per_s": nnz / (valu_step_us * 1e-6),
                          "frac_fwd": round(attn_bytes["gt_hyper_fwd"] / (valu_us["gt_hyper_fwd"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                          "frac_bwd": round(attn_bytes["gt_bwd"] / (valu_us["gt_bwd"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}}

    # ---- secondary figures (never part of `value`), rank 0 at N = 1 only ----------------------------------------
    secondary = None
    if rank == 0 and world == 1:
        import dfgnn_preprocess
        import fused_gatconv
        from _binding_util import build_plan
        from DFGNN.utils import GraphedStep
        del out, attn, gedge, dQ, dK, dV
        torch.cuda.empty_cache()
        with torch.no_grad():
            # the GAT training pair through the C ABI with preallocated outputs, like the timed launches above (through the
            # binding the per-call output allocations sometimes miss the allocator's cache and the events then bracket host
            # time: 117 instead of 60 us)
            ar, ac, X = S.gat_features(m, h, f, seed=6, device=dev)
            g_rows, g_plan, g_meta = fused_gatconv._train_plan(W.row_ptr, W.col_ind, f, 0.0)
            go, gfeat = torch.empty_like(X), torch.empty_like(X)
            emax, esum, g_ar, g_ac = (torch.empty(m, h, device=dev) for _ in range(4))
            g_edge = torch.empty(h, nnz, device=dev)
            gat_f = ev_us(checked(lambda: L.dfgnn_gat_fwd_train(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(g_rows), P(ar), P(ac),
                                                                0.2, P(X), None, 0.0, P(emax), P(esum), P(go), g_plan, g_meta,
                                                                stream)))
            gat_b = ev_us(checked(lambda: L.dfgnn_gat_bwd(m, nnz, h, f, P(W.row_ptr), P(W.col_ind), P(g_rows), P(W.col_ptr),
                                                          P(W.row_ind), P(W.val_idx), P(ar), P(ac), 0.2, P(X), P(emax), P(esum),
                                                          None, 0.0, P(W.dO), P(g_edge), P(gfeat), P(g_ar), P(g_ac), g_plan,
                                                          g_meta, stream)))
        src, dst = W.g.edges()

        def raw_step():  # the launches of the timed step as explicit operator calls (no autograd bookkeeping)
            with torch.no_grad():
                return fused_gtconv.gt_hyper_step_raw(W.row_ptr, W.col_ind, W.rows, W.val, W.col_ptr, W.row_ind, W.val_idx,
                                                      W.smem, W.Q, W.K, W.V, W.dO)[1:]

        graphed = GraphedStep(raw_step)
        secondary = {
            # the reference's form of the pair (attn_edge in CSR order written by the forward, read by the backward), same
            # batch, same launches as round 2's headline; SURVEY.md 8(d) bytes
            "attn_edge_pair": {k: {"avg_us": round(v, 2), "algorithmic_bytes": attn_bytes[k],
                                   "frac": round(attn_bytes[k] / (v * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)} for k, v in attn_us.items()},
            # the pair that saves row statistics instead of attn_edge (what the step launches from two heads on)
            "weighted_edges": weighted,
            "row_statistics_pair": None if stats_us is None else {
                k: {"avg_us": round(v, 2), "algorithmic_bytes": stats_bytes[k],
                    "frac": round(stats_bytes[k] / (v * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)} for k, v in stats_us.items()},
            "gat_train": {"fwd_us": round(gat_f, 1), "bwd_us": round(gat_b, 1),
                          "edges_per_s": nnz / ((gat_f + gat_b) * 1e-6)},
            "preprocess_per_batch_ms": {
                "coo_to_csr_csc_native": round(wall_ms(lambda: dfgnn_preprocess.coo_to_hyper(src, dst, m, csc=True)), 3),
                "block_plan": round(wall_ms(lambda: build_plan(W.row_ptr, W.col_ind, f)), 3)},
            "step_as_hipgraph_ms": round(wall_ms(graphed.replay, reps=max(10, args.steps)), 4),
        }
        if not args.no_c4:
            secondary["c4"] = bench_c4(dev)

    # ---- CPU baseline: the oracle's fp32 OpenMP build on a bounded sample of the same workload -------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # (the contract: rank 0 at N = 1 only)
        import oracle
        oracle.build()
        ng = min(args.cpu_sample_graphs, args.batch_size)
        sizes = W.g_host.batch_num_nodes().numpy()
        n_s = int(sizes[:ng].sum())
        rp = W.row_ptr[: n_s + 1].cpu().numpy()
        e_s = int(rp[-1])
        ci = W.col_ind[:e_s].cpu().numpy()
        vl = W.val[:e_s].cpu().numpy()
        q, k, v, do = (t[:n_s].detach().cpu().numpy() for t in (W.Q, W.K, W.V, W.dO))
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
        # BASELINE.json configs[0] (the reference's own CPU-runnable case): GAT on the cora-like graph, f = 64, CSR -- the
        # same CPU port, timed here because this leg is the one place of the bench that may run the oracle
        from DFGNN.layers import preprocess_CSR
        gc = S.cora_like()
        c_rp, c_ci, _, _ = preprocess_CSR(gc)
        c_ar, c_ac, c_x = S.gat_features(gc.num_nodes(), 1, 64, seed=4)
        c_args = [t.numpy() for t in (c_rp, c_ci, c_ar, c_ac)] + [0.2, c_x.numpy()]
        for _ in range(3):
            oracle.gat_forward(*c_args, acc="f32")
        c_times = []
        for _ in range(20):   # median of the calls: a 55 us call on 128 OpenMP threads is at the mercy of the host's other tenants
            t0 = time.perf_counter()
            oracle.gat_forward(*c_args, acc="f32")
            c_times.append(time.perf_counter() - t0)
        c_dt = sorted(c_times)[len(c_times) // 2]
        cpu["c1_gat_cora_f64"] = {"workload": "GAT conv 'csr' on a cora-like graph, dim=64 (BASELINE.json configs[0]), CPU port",
                                  "edges": gc.num_edges(), "ms": round(c_dt * 1e3, 4),
                                  "edges_per_s": gc.num_edges() / c_dt, "cores": oracle.num_threads()}

    if rank == 0:
        line = {
            "metric": "edges/s (fwd+bwd) fused GT conv, PATTERN bs=1024 dim=128",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "setup_steps": SETUP_STEPS,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None, "dtype": "f32", "arithmetic": arithmetic, "data": "synthetic",
            "config": {"workload": f"GT conv 'hyper' fwd+bwd, PATTERN-like batch bs={args.batch_size} "
                                   f"dim={args.dim} heads={h} (BASELINE.json configs[2])",
                       "training_pair": "row statistics (dfgnn_gt_hyper_fwd_stats / dfgnn_gt_bwd_stats)" if uses_stats is not None
                       else "attn_edge in rank order (dfgnn_gt_hyper_fwd_ranked / dfgnn_gt_bwd_ranked)" if ranked_us is not None
                       else "attn_edge (dfgnn_gt_hyper_fwd / dfgnn_gt_bwd)",
                       "nodes_per_gpu": m, "edges_per_gpu": nnz, "total_edges": total_edges,
                       "parallelism": (f"graph-sharded x{world} ({args.scaling} scaling), no data-path collective")},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if other:
            line["strong_scaling" if args.scaling == "weak" else "weak_scaling"] = other
        if gather:
            line["inference_allgather"] = gather
        if secondary:
            line["secondary"] = secondary
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def bench_c4(dev):
    """BASELINE.json configs[3]: GAT 'tiling' on the reddit-like super-node graph (232 965 nodes, 114.6 M edges, f = 128),
    single GPU; edges/s, achieved-HBM fraction (algorithmic bytes) and the row-gather rate that physically bounds it."""
    from DFGNN.layers.util import preprocess_CSR
    from DFGNN.operators import fused_gatconv as gat
    from DFGNN.utils import synthetic as S
    t0 = time.perf_counter()
    g = S.reddit_like().to(dev)
    row_ptr, col_ind, val, _ = preprocess_CSR(g)
    m, nnz = g.num_nodes(), g.num_edges()
    build_s = time.perf_counter() - t0
    ar, ac, X = S.gat_features(m, 1, 128, seed=4, device=dev)
    import fused_gatconv as fg
    chunked = fg._use_chunked_tiling(m, nnz, 1, 128)
    t1 = time.perf_counter()
    gat.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X)   # first call: builds the chunk-major edge order (cached)
    torch.cuda.synchronize()
    first_ms = (time.perf_counter() - t1) * 1e3
    us = ev_us(lambda: gat.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X), reps=5, warm=2)
    keep = fg.TILING_CHUNK_ROWS
    try:  # the single-kernel form (a wave per row, gathers served by the Infinity Cache) for comparison
        fg.TILING_CHUNK_ROWS = 0
        us_single = ev_us(lambda: gat.GATConvFuse_inference_tiling(ar, ac, row_ptr, col_ind, 0.2, X), reps=5, warm=2)
    finally:
        fg.TILING_CHUNK_ROWS = keep
    byt = 8 * m * 128 + 8 * m + 4 * (m + 1) + 4 * nnz
    deg = row_ptr[1:] - row_ptr[:-1]
    res = {"workload": "GAT conv 'tiling' on a reddit-like graph, dim=128 (BASELINE.json configs[3])", "nodes": m,
           "edges": nnz, "max_degree": int(deg.max()), "ms": round(us / 1e3, 3), "edges_per_s": nnz / (us * 1e-6),
           "algorithmic_bytes": byt, "hbm_frac": round(byt / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
           "gather_GBs": round(nnz * 512 / (us * 1e-6) / 1e9, 1),
           "variant": ("column chunks of %d rows pinned per XCD + partial-state merge (csrc/gat_tiling_chunked.hip)"
                       % fg._chunk_rows(128)) if chunked else "single kernel, a wave per row",
           "single_kernel_ms": round(us_single / 1e3, 3), "first_call_ms_incl_chunk_build": round(first_ms, 1),
           "graph_build_s": round(build_s, 1)}
    del g, row_ptr, col_ind, val, ar, ac, X
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
