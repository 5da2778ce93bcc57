"""Host-side mirror of NNop's RMSNorm and LayerNorm operators (src/rms_norm.jl, src/layer_norm.jl) over the C ABI.

    y = rms_norm(x, w, eps=1e-6, offset=0.0)                 # rms_norm.jl:171-176, differentiable (rrule :178-185)
    y, rms = _rms_norm(x, w, eps=..., offset=...)            # :117-137
    dx, dw = grad_rms_norm(dy, rms, x, w, offset=...)        # ∇rms_norm :139-169 (dw is float32, :146)
    y = layer_norm(x, w, b, eps=1e-6)                        # layer_norm.jl:206-211 (rrule :213-220)
    y, mu, sigma = _layer_norm(x, w, b, eps=...)             # :150-170
    dx, dw, db = grad_layer_norm(dy, mu, sigma, x, w, b)     # ∇layer_norm :172-204

Layout: x [n, emb] == Julia (emb, n); w, b [emb] in float32 or the dtype of x.  GPU-only like the reference kernels.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import NormDesc
from .attention import NNopError, _DTYPES, _on_device, _ptr, _stream

__all__ = ["rms_norm", "_rms_norm", "grad_rms_norm", "layer_norm", "_layer_norm", "grad_layer_norm"]


def _check(x, w, b=None):
    if not isinstance(x, torch.Tensor) or x.dim() != 2:
        raise TypeError("`x` must be a matrix [n, emb]")
    if not x.is_cuda:
        raise NNopError("NNop norms are GPU-only: tensors must live on a HIP device "
                        "(there is no CPU or PyTorch fallback).")
    if x.dtype not in _DTYPES:
        raise TypeError(f"unsupported element type {x.dtype}; expected float32, float16 or bfloat16")
    if x.shape[0] == 0 or x.shape[1] == 0:
        raise NNopError("norms need a non-empty matrix")
    for name, t in (("w", w), ("b", b)):
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or t.dim() != 1 or t.device != x.device:
            raise TypeError(f"`{name}` must be a vector on the device of x")
        if t.shape[0] != x.shape[1]:
            # the reference: @assert emb == length(w), src/rms_norm.jl:119
            raise NNopError(f"AssertionError: emb == length({name}) ({x.shape[1]} vs {t.shape[0]})")
        if t.dtype not in (torch.float32, x.dtype):
            raise TypeError(f"`{name}` must be float32 or share the dtype of x")
    if b is not None and b.dtype != w.dtype:
        raise TypeError("w and b must share one dtype")


def _desc(x, w):
    return NormDesc(dtype=_DTYPES[x.dtype], w_dtype=_DTYPES[w.dtype], emb=x.shape[1], reserved=0, n=x.shape[0])


def _raise(st):
    if st != _lib.NNOP_OK:
        raise NNopError(_lib.strerror(st), st)


def _workspace(d, ln, device):
    nbytes = int(_lib.load().nnop_norm_bwd_workspace_bytes(C.byref(d), 1 if ln else 0))
    return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device), nbytes


def _rms_norm(x, w, *, eps: float = 1e-6, offset: float = 0.0):
    """``NNop._rms_norm(x, w; ϵ, offset)`` -> (y, rms)."""
    _check(x, w)
    x, w = x.contiguous(), w.contiguous()
    y = torch.empty_like(x)
    rms = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    with _on_device(x):
        _raise(_lib.load().nnop_rms_norm(C.byref(_desc(x, w)), _ptr(y), _ptr(rms), _ptr(x), _ptr(w),
                                         C.c_float(offset), C.c_float(eps), _stream(x)))
    return y, rms


def grad_rms_norm(dy, rms, x, w, *, offset: float = 0.0):
    """``∇rms_norm(Δ, rms, x, w; offset)`` -> (dx, dw) with dw in float32."""
    _check(x, w)
    if dy.shape != x.shape or dy.dtype != x.dtype or dy.device != x.device:
        raise TypeError("Δ must match x in shape, dtype and device")
    if rms.dtype != torch.float32 or tuple(rms.shape) != (x.shape[0],):
        raise TypeError("rms must be a float32 vector [n]")
    dy, x, w, rms = dy.contiguous(), x.contiguous(), w.contiguous(), rms.contiguous()
    d = _desc(x, w)
    dx = torch.empty_like(x)
    dw = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    ws, nbytes = _workspace(d, False, x.device)
    with _on_device(x):
        _raise(_lib.load().nnop_rms_norm_bwd(C.byref(d), _ptr(dx), _ptr(dw), _ptr(dy), _ptr(rms), _ptr(x), _ptr(w),
                                             C.c_float(offset), _ptr(ws), C.c_size_t(nbytes), _stream(x)))
    return dx, dw


def _layer_norm(x, w, b, *, eps: float = 1e-6):
    """``NNop._layer_norm(x, w, b; ϵ)`` -> (y, μ, Σ)."""
    _check(x, w, b)
    x, w, b = x.contiguous(), w.contiguous(), b.contiguous()
    y = torch.empty_like(x)
    mu = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    sigma = torch.empty_like(mu)
    with _on_device(x):
        _raise(_lib.load().nnop_layer_norm(C.byref(_desc(x, w)), _ptr(y), _ptr(mu), _ptr(sigma), _ptr(x), _ptr(w),
                                           _ptr(b), C.c_float(eps), _stream(x)))
    return y, mu, sigma


def grad_layer_norm(dy, mu, sigma, x, w, b=None):
    """``∇layer_norm(Δ, μ, Σ, x, w, b)`` -> (dx, dw, db); dw, db in the dtype of w (src/layer_norm.jl:179-180)."""
    _check(x, w, b)
    if dy.shape != x.shape or dy.dtype != x.dtype or dy.device != x.device:
        raise TypeError("Δ must match x in shape, dtype and device")
    for s in (mu, sigma):
        if s.dtype != torch.float32 or tuple(s.shape) != (x.shape[0],):
            raise TypeError("μ and Σ must be float32 vectors [n]")
    dy, x, w, mu, sigma = dy.contiguous(), x.contiguous(), w.contiguous(), mu.contiguous(), sigma.contiguous()
    d = _desc(x, w)
    dx = torch.empty_like(x)
    dw, db = torch.empty_like(w), torch.empty_like(w)
    ws, nbytes = _workspace(d, True, x.device)
    with _on_device(x):
        _raise(_lib.load().nnop_layer_norm_bwd(C.byref(d), _ptr(dx), _ptr(dw), _ptr(db), _ptr(dy), _ptr(mu), _ptr(sigma),
                                               _ptr(x), _ptr(w), _ptr(ws), C.c_size_t(nbytes), _stream(x)))
    return dx, dw, db


class _RMSNorm(torch.autograd.Function):
    """rrule of src/rms_norm.jl:178-185."""

    @staticmethod
    def forward(ctx, x, w, eps, offset):
        y, rms = _rms_norm(x, w, eps=eps, offset=offset)
        ctx.save_for_backward(rms, x, w)
        ctx.offset = offset
        return y

    @staticmethod
    def backward(ctx, dy):
        rms, x, w = ctx.saved_tensors
        dx, dw = grad_rms_norm(dy, rms, x, w, offset=ctx.offset)
        return dx, dw.to(w.dtype), None, None


class _LayerNorm(torch.autograd.Function):
    """rrule of src/layer_norm.jl:213-220."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        y, mu, sigma = _layer_norm(x, w, b, eps=eps)
        ctx.save_for_backward(mu, sigma, x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        mu, sigma, x, w = ctx.saved_tensors
        dx, dw, db = grad_layer_norm(dy, mu, sigma, x, w)
        return dx, dw, db, None


def rms_norm(x, w, *, eps: float = 1e-6, offset: float = 0.0):
    """``NNop.rms_norm(x, w; ϵ=1f-6, offset=0f0)`` (src/rms_norm.jl:171-176)."""
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad):
        _check(x, w)
        return _RMSNorm.apply(x, w, float(eps), float(offset))
    return _rms_norm(x, w, eps=eps, offset=offset)[0]


def layer_norm(x, w, b, *, eps: float = 1e-6):
    """``NNop.layer_norm(x, w, b; ϵ=1f-6)`` (src/layer_norm.jl:206-211)."""
    if torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or b.requires_grad):
        _check(x, w, b)
        return _LayerNorm.apply(x, w, b, float(eps))
    return _layer_norm(x, w, b, eps=eps)[0]
