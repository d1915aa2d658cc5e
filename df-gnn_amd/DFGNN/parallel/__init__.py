"""Multi-GPU use of the fused convolution: batched datasets shard by whole graphs.
The reference has no distributed code at all (single cuda:0); this is new, MI355X-side design:
one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm), gloo on CPU for tests."""
from .sharding import (ShardedGTConv, all_gather_rows, shard_graph, shard_graph_bounds,  # noqa: F401
                       shard_rows)
from .overlap import OverlappedGradSync  # noqa: F401
