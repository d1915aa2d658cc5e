"""Gradient exchange overlapped with the backward pass, for data-parallel training over graph shards.

The fused convolution needs no collective (block-diagonal batch => dQ / dK / dV of a graph stay on the rank that owns
it, DFGNN/parallel/sharding.py); what a multi-GPU training step of the reference's 8-layer stack
(DFGNN/script/train/train_batch_graph_timing.py:32-53, 146-196) does exchange is the WEIGHT gradients.  The autograd
engine produces them last layer first, so each layer's bucket can travel over xGMI while the layers before it are still
in their backward kernels: one `all_reduce(async_op=True)` per bucket (RCCL runs it on its own stream), issued from a
post-accumulate-grad hook as soon as the bucket's last gradient exists, waited for once before the optimizer step.
Buckets are per top-level block of the model (one fused conv layer = its q/k/v projections): a few hundred KB each --
latency-bound on a ring, which is why they are issued early instead of being merged into one late transfer.
"""
import torch
import torch.distributed as dist


class OverlappedGradSync:
    """sync = OverlappedGradSync(model); loss.backward(); sync.finish(); optimizer.step()

    Gradients are SUMMED over the ranks (scale the loss by the global sample count for a mean).  `overlap=False` does
    the same exchange in finish(), after the whole backward -- the baseline the overlap is measured against."""

    def __init__(self, model, group=None, overlap=True):
        self.group, self.overlap = group, overlap
        self.buckets = []                       # [params], flat buffer, ready count, handle
        by_block = {}
        for name, p in model.named_parameters():
            if p.requires_grad:
                block = name.split(".")[0] if "." not in name or not name.split(".")[1].isdigit() else ".".join(name.split(".")[:2])
                by_block.setdefault(block, []).append(p)
        for block, params in by_block.items():
            flat = torch.zeros(sum(p.numel() for p in params), dtype=params[0].dtype, device=params[0].device)
            bucket = {"name": block, "params": params, "flat": flat, "ready": 0, "handle": None}
            self.buckets.append(bucket)
            off = 0
            for p in params:
                p.register_post_accumulate_grad_hook(self._hook(bucket, off, p.numel()))
                off += p.numel()
        self._next = len(self.buckets) - 1      # the bucket whose all-reduce goes out next

    def _issue_ready(self, everything=False):
        # collectives go out in ONE order on every rank -- last block first, the order the backward fills them in -- and
        # a bucket waits for the ones after it: ranks that fill different buckets still pair up the same calls
        while self._next >= 0 and (everything or self.buckets[self._next]["ready"] == len(self.buckets[self._next]["params"])):
            b = self.buckets[self._next]
            b["handle"] = dist.all_reduce(b["flat"], group=self.group, async_op=True)
            self._next -= 1

    def _hook(self, bucket, off, n):
        def fn(p):
            if bucket["handle"] is not None:
                raise RuntimeError("OverlappedGradSync: a second backward reached bucket '%s' while its all-reduce from "
                                   "the previous one is still pending -- call finish() after every backward" % bucket["name"])
            bucket["flat"][off:off + n].copy_(p.grad.reshape(-1))
            bucket["ready"] += 1
            if self.overlap:
                self._issue_ready()
        return fn

    def finish(self):
        """Wait for every bucket (issue the ones that were not overlapped) and write the summed gradients back.
        EVERY bucket is reduced on EVERY rank, in the same order, whether or not a gradient reached it here (its slots
        are then zeros): ranks whose shards exercise different parameters still issue the same collectives.  A
        parameter without a local gradient receives the other ranks' sum."""
        self._issue_ready(everything=True)
        self._next = len(self.buckets) - 1
        for b in self.buckets:
            b["handle"].wait()
            off = 0
            for p in b["params"]:
                piece = b["flat"][off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    p.grad.copy_(piece)
                else:
                    p.grad = piece.clone()
                off += p.numel()
            b["flat"].zero_()                   # slots no gradient reaches in the next backward travel as zeros
            b["ready"], b["handle"] = 0, None
