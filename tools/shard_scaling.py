#!/usr/bin/env python3
"""Strong scaling of the headline step, PROJECTED from one GPU (the pool gives one GPU per call; the driver measures the
real curve when it has an 8-GPU node): the seed-1 bs=1024 PATTERN-like batch is cut into G shards of whole graphs
(DFGNN/parallel/sharding.py, edge-balanced) and every shard's fwd+bwd step is timed on this GPU.  The step has no
data-path collective (block-diagonal batch), so G GPUs finish when the slowest shard does: t(G) = max over shards.
Prints one JSON line per G: eager wall time per step and the same launches replayed as one HIP graph.
usage: python3 tools/shard_scaling.py [--shards 1,2,4,8] [--steps 30]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "df-gnn_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402  (Workload, wall_ms)
import fused_gtconv  # noqa: E402
from DFGNN.parallel import shard_graph  # noqa: E402
from DFGNN.utils import GraphedStep  # noqa: E402
from DFGNN.utils import synthetic as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch-size", type=int, default=1024)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    full = S.pattern_like(batch_size=args.batch_size, seed=1)
    total_edges = full.num_edges()
    base = None
    for G in [int(x) for x in args.shards.split(",")]:
        eager, graph, sizes = [], [], []
        for r in range(G):
            sub = full if G == 1 else shard_graph(full, r, G)[0]
            W = bench.Workload(sub, 1, 128, 100 + r, dev)

            def raw_step(W=W):
                with torch.no_grad():
                    return fused_gtconv.gt_hyper_step_raw(W.row_ptr, W.col_ind, W.rows, W.val, W.col_ptr, W.row_ind,
                                                          W.val_idx, W.smem, W.Q, W.K, W.V, W.dO)[1:]

            # median of five windows: one host hiccup inside a window (allocator, collector) otherwise shows up as a
            # 10x outlier of a ~0.1 ms step
            med = lambda fn: sorted(bench.wall_ms(fn, reps=args.steps, warm=5) for _ in range(5))[2]  # noqa: E731
            eager.append(med(W.step))
            graph.append(med(GraphedStep(raw_step).replay))
            sizes.append((len(sub.batch_num_nodes()), W.nnz))
            del W
            torch.cuda.empty_cache()
        t_e, t_g = max(eager), max(graph)
        if base is None:
            base = (t_e * G, t_g * G) if G == 1 else None
        line = {"shards": G, "graphs_per_shard": [s[0] for s in sizes][:4], "edges_per_shard_max": max(s[1] for s in sizes),
                "ms_per_step_eager_max_over_shards": round(t_e, 4), "ms_per_step_hipgraph_max_over_shards": round(t_g, 4),
                "projected_edges_per_s_eager": total_edges / (t_e * 1e-3), "projected_edges_per_s_hipgraph": total_edges / (t_g * 1e-3)}
        if base:
            line["efficiency_eager"] = round(base[0] / (G * t_e), 3)
            line["efficiency_hipgraph"] = round(base[1] / (G * t_g), 3)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
