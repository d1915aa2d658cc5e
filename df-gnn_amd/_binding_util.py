"""Checks shared by the fused_gtconv / fused_gatconv binding modules.

Mirrors the reference's binding-level checks (DFGNN/src/fused_gtconv/fused_gtconv.cpp:7-13:
CHECK_DEVICE / CHECK_CONTIGUOUS raise RuntimeError) and turns its compiled-out dtype asserts
(fused_gtconv.cpp:103-112) into real errors.
"""
import torch


def check_device(**tensors):
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be on CUDA")


def check_contiguous(**tensors):
    for name, t in tensors.items():
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")


def check_dtype(dtype, **tensors):
    for name, t in tensors.items():
        if t.dtype != dtype:
            raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")


def as_int32(t):
    """Index arrays are int32 on the device; int64 (e.g. dgl's val_idx) is narrowed once here."""
    return t if t.dtype == torch.int32 else t.to(torch.int32)


def check_feat3(**tensors):
    shape = None
    for name, t in tensors.items():
        if t.dim() != 3:
            raise RuntimeError(f"{name} must have shape [nodes, heads, feat], got {tuple(t.shape)}")
        if shape is None:
            shape = t.shape
        elif t.shape != shape:
            raise RuntimeError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")


def stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    return t.data_ptr() if t is not None else None
