"""HIP-graph capture of an operator step (MI355X-specific helper; no counterpart in the reference).

Small batches (Peptides-like: ~70 us of kernels per fwd+bwd) are bound by the host side of each launch -- the
Python binding, autograd bookkeeping and the launch itself cost more than the kernels.  Every entry point of the C ABI
is capturable (no allocation, no host synchronisation inside, launches on the caller's stream), so a whole training
step can be recorded once and replayed with a single hipGraphLaunch:

    def fwd_bwd():                                        # explicit operator calls (fused_gtconv / fused_gatconv)
        return fused_gtconv.gt_hyper_step_raw(row_ptr, col_ind, rows, val, col_ptr, row_ind, val_idx, smem, Q, K, V, dO)
    step = GraphedStep(fwd_bwd)                           # warm-up (builds and caches the block plan), then capture
    out, dQ, dK, dV = step.replay()                       # same tensors every time: copy new inputs in place

Record operator calls, not `autograd.Function.apply` + `torch.autograd.grad` / `.backward()`.  A callable that runs the
autograd engine inside the capture took the process down on this ROCm / torch build (round 1: `Fatal Python error:
Segmentation fault` in torch/cuda/graphs.py capture_end <- hipStreamEndCapture, a second native thread -- the engine's
device worker -- alive, preceded by torch's "AccumulateGrad node's stream does not match ... may break CUDA graph
capture" warning).  Cause: the engine replays each node on the stream it was first recorded on; the leaves' AccumulateGrad
nodes were created during the warm-up on another stream, so the captured backward forks onto that stream through event
waits issued from the worker thread and nothing joins it back before hipStreamEndCapture -- unjoined cross-stream,
cross-thread work in a global-mode capture, which this runtime answers with a crash instead of
hipErrorStreamCaptureUnjoined (intermittently: it depends on which thread's launch lands first).  GraphedStep therefore
runs the callable with autograd switched off: operator calls are unaffected, an autograd step fails up front with a
RuntimeError and nothing is captured.

The inputs are static tensors: write the next batch's features into them (`Q.copy_(...)`) before `replay()`.  The graph
structure (index arrays, plan) is baked in, i.e. one GraphedStep per batch structure -- the case of multi-layer /
multi-epoch training over cached batches (DFGNN/script/train/train_batch_graph_timing.py keeps its batches on the GPU).
"""
import torch


def _refuse_autograd(outputs):
    flat = outputs if isinstance(outputs, (list, tuple)) else [outputs]
    for t in flat:
        if isinstance(t, torch.Tensor) and (t.grad_fn is not None or t.requires_grad):
            raise RuntimeError("GraphedStep: the callable returned a tensor that records autograd; capture explicit "
                               "operator calls (fused_gtconv.gt_hyper_forward / gt_backward, ...) instead")


class GraphedStep:
    def __init__(self, fn, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.no_grad(), torch.cuda.stream(side):  # warm-up off the default stream: plan build, allocator pools
                for _ in range(warmup):
                    _refuse_autograd(fn())
        except RuntimeError as e:
            if "does not require grad" in str(e) or "GraphedStep" in str(e):
                raise RuntimeError(
                    "GraphedStep captures explicit operator calls only: the callable runs the autograd engine "
                    "(autograd.Function.apply + torch.autograd.grad / .backward()), whose worker thread and per-node "
                    "streams cannot be captured safely (DFGNN/utils/hipgraph.py)") from e
            raise
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs
