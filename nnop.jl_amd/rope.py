"""Host-side mirror of NNop.jl's Llama RoPE operator (src/rope/llama_rope.jl) over the C ABI.

    emb = LlamaRotaryEmbedding(dim, base=10000)              # :1-12
    cos, sin = emb(position_ids)                              # :15-22   position_ids [B, L] -> cos, sin [B, L, dim]
    q2, k2 = llama_rope(q, k, cos=cos, sin=sin)               # :67      differentiable (rrule :94-98)
    dq, dk = grad_llama_rope((dq2, dk2), cos, sin)            # :91-92   ∇llama_rope

Layout: q [B, QH, L, D], k [B, KH, L, D], cos/sin [B, L, D] -- the same memory as the reference's Julia arrays
(D, L, H, B) / (D, L, B).  GPU-only like the reference kernel (`cpu=false`, :24): no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import RopeDesc
from .attention import NNopError, _DTYPES, _on_device, _ptr, _stream

__all__ = ["LlamaRotaryEmbedding", "llama_rope", "_llama_rope", "grad_llama_rope", "llama_rope_into"]


class LlamaRotaryEmbedding:
    """``LlamaRotaryEmbedding(dim; base=10000)`` (src/rope/llama_rope.jl:1-12): holds ``inv_freq`` (fp32, dim/2).

    Calling it with ``position_ids`` [B, L] (the reference: (L, B) Float32) returns ``(cos, sin)`` [B, L, dim] fp32
    built from ``vcat(freqs, freqs)`` (:15-22).  This is table set-up (a broadcast in the reference too), not a kernel
    of the operator; it runs as device array ops on the device of ``position_ids``.
    """

    def __init__(self, dim: int, base: int = 10000, device=None):
        self.dim = int(dim)
        ids = torch.arange(0, self.dim, 2, dtype=torch.float32, device=device) / float(self.dim)
        self.inv_freq = 1.0 / (torch.tensor(float(base), dtype=torch.float32, device=device) ** ids)

    def to(self, device):
        self.inv_freq = self.inv_freq.to(device)
        return self

    def __call__(self, position_ids):
        pos = position_ids.to(torch.float32)
        inv = self.inv_freq.to(pos.device)
        freqs = pos[..., None] * inv                          # [B, L, dim/2]
        freqs = torch.cat([freqs, freqs], dim=-1)
        return torch.cos(freqs), torch.sin(freqs)


def _check(q, k, cos, sin):
    for name, t in (("q", q), ("k", k)):
        if not isinstance(t, torch.Tensor) or t.dim() != 4:
            raise TypeError(f"`{name}` must be a 4-D tensor [B, H, L, D]")
    if not q.is_cuda:
        raise NNopError("NNop llama_rope is GPU-only: tensors must live on a HIP device "
                        "(there is no CPU or PyTorch fallback).")
    if q.dtype not in _DTYPES:
        raise TypeError(f"unsupported element type {q.dtype}; expected float32, float16 or bfloat16")
    if k.dtype != q.dtype or k.device != q.device:
        raise TypeError("q and k must share one dtype and device")
    B, QH, L, D = q.shape
    # the reference's @assert lines, src/rope/llama_rope.jl:72-73
    if k.shape[3] != D or k.shape[2] != L or k.shape[0] != B:
        raise NNopError(f"AssertionError: q {tuple(q.shape)} and k {tuple(k.shape)} must agree in head dim, "
                        "sequence length and batch.")
    for name, t in (("cos", cos), ("sin", sin)):
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != (B, L, D) or t.device != q.device:
            raise NNopError(f"`{name}` must have shape [B, L, D] = {(B, L, D)} on the device of q")
    if cos.dtype != sin.dtype or cos.dtype not in (torch.float32, q.dtype):
        raise TypeError("cos and sin must share one dtype: float32 or the dtype of q")
    if D % 2:
        raise NNopError("head dim must be even")


def llama_rope_into(q_out, k_out, q, k, cos, sin, *, bwd: bool = False):
    """Lowest-level call: writes into caller-owned buffers (q_out may be q, k_out may be k)."""
    _check(q, k, cos, sin)
    q, k, cos, sin = (t if t.is_contiguous() else t.contiguous() for t in (q, k, cos, sin))
    for o, x in ((q_out, q), (k_out, k)):
        if o.shape != x.shape or o.dtype != x.dtype or o.device != x.device or not o.is_contiguous():
            raise NNopError("output buffers must be dense and match q / k in shape, dtype and device")
    B, QH, L, D = q.shape
    d = RopeDesc(dtype=_DTYPES[q.dtype], cs_dtype=_DTYPES[cos.dtype], dim=D, seq=L, qh=QH, kh=k.shape[1], batch=B)
    with _on_device(q):
        st = _lib.load().nnop_llama_rope(C.byref(d), _ptr(q_out), _ptr(k_out), _ptr(q), _ptr(k), _ptr(cos), _ptr(sin),
                                         C.c_float(-1.0 if bwd else 1.0), _stream(q))
    if st != _lib.NNOP_OK:
        raise NNopError(_lib.strerror(st), st)
    return q_out, k_out


def _llama_rope(q, k, cos, sin, *, bwd: bool):
    """``NNop._llama_rope(q, k, cos, sin; bwd)`` (src/rope/llama_rope.jl:69-89): returns rotated COPIES."""
    return llama_rope_into(torch.empty_like(q, memory_format=torch.contiguous_format),
                           torch.empty_like(k, memory_format=torch.contiguous_format), q, k, cos, sin, bwd=bwd)


def grad_llama_rope(grads, cos, sin):
    """``∇llama_rope((dq, dk), cos, sin)`` (src/rope/llama_rope.jl:91-92)."""
    dq, dk = grads
    return _llama_rope(dq, dk, cos, sin, bwd=True)


class _LlamaRope(torch.autograd.Function):
    """The rrule of src/rope/llama_rope.jl:94-98: q, k get ∇llama_rope; cos, sin get no tangent."""

    @staticmethod
    def forward(ctx, q, k, cos, sin):
        ctx.save_for_backward(cos, sin)
        return _llama_rope(q, k, cos, sin, bwd=False)

    @staticmethod
    def backward(ctx, dq, dk):
        cos, sin = ctx.saved_tensors
        gq, gk = grad_llama_rope((dq.contiguous(), dk.contiguous()), cos, sin)
        return gq, gk, None, None


def llama_rope(q, k, *, cos, sin):
    """``llama_rope(q, k; cos, sin)`` (src/rope/llama_rope.jl:67)."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad):
        return _LlamaRope.apply(q, k, cos, sin)
    return _llama_rope(q, k, cos, sin, bwd=False)
