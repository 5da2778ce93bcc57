"""Host-side mirror of NNop.online_softmax (src/softmax.jl:60-86) over the C ABI.

    y  = online_softmax(x)                   # x [batch, N] == Julia (N, batch); softmax along N; differentiable
    dx = grad_online_softmax(dy, y)          # ∇online_softmax(Δ, y), :70-80

GPU-only like the reference kernel (`cpu=false`, :19): no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SoftmaxDesc
from .attention import NNopError, _DTYPES, _on_device, _ptr, _stream

__all__ = ["online_softmax", "grad_online_softmax", "online_softmax_into"]


def _check(x, name="x"):
    if not isinstance(x, torch.Tensor) or x.dim() != 2:
        # the reference's signature: x::AbstractMatrix (src/softmax.jl:60) -> MethodError otherwise
        raise TypeError(f"`{name}` must be a matrix [batch, N]")
    if not x.is_cuda:
        raise NNopError("NNop online_softmax is GPU-only: tensors must live on a HIP device "
                        "(there is no CPU or PyTorch fallback).")
    if x.dtype not in _DTYPES:
        raise TypeError(f"unsupported element type {x.dtype}; expected float32, float16 or bfloat16")
    if x.shape[0] == 0 or x.shape[1] == 0:
        raise NNopError("online_softmax needs a non-empty matrix")


def _desc(x):
    return SoftmaxDesc(dtype=_DTYPES[x.dtype], n=x.shape[1], batch=x.shape[0])


def online_softmax_into(y, x):
    """Lowest-level call: writes into a caller-owned buffer (y may alias x)."""
    _check(x)
    x = x if x.is_contiguous() else x.contiguous()
    if y.shape != x.shape or y.dtype != x.dtype or y.device != x.device or not y.is_contiguous():
        raise NNopError("output buffer must be dense and match x in shape, dtype and device")
    with _on_device(x):
        st = _lib.load().nnop_online_softmax(C.byref(_desc(x)), _ptr(y), _ptr(x), _stream(x))
    if st != _lib.NNOP_OK:
        raise NNopError(_lib.strerror(st), st)
    return y


def grad_online_softmax(dy, y):
    """``∇online_softmax(Δ, y)`` (src/softmax.jl:70-80), fused into one pass."""
    _check(y, "y")
    if not isinstance(dy, torch.Tensor) or dy.shape != y.shape or dy.dtype != y.dtype or dy.device != y.device:
        raise TypeError("Δ must match y in shape, dtype and device")
    dy, y = dy.contiguous(), y.contiguous()
    dx = torch.empty_like(y)
    with _on_device(y):
        st = _lib.load().nnop_online_softmax_bwd(C.byref(_desc(y)), _ptr(dx), _ptr(dy), _ptr(y), _stream(y))
    if st != _lib.NNOP_OK:
        raise NNopError(_lib.strerror(st), st)
    return dx


class _OnlineSoftmax(torch.autograd.Function):
    """The rrule of src/softmax.jl:82-86."""

    @staticmethod
    def forward(ctx, x):
        y = online_softmax_into(torch.empty_like(x, memory_format=torch.contiguous_format), x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return grad_online_softmax(dy, y)


def online_softmax(x):
    """``NNop.online_softmax(x)`` (src/softmax.jl:60-68)."""
    _check(x)
    if torch.is_grad_enabled() and x.requires_grad:
        return _OnlineSoftmax.apply(x)
    return online_softmax_into(torch.empty_like(x, memory_format=torch.contiguous_format), x)
