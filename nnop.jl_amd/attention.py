"""Host-side mirror of NNop.jl's Flash-Attention operator interface over the C ABI.

Same names, argument meaning and error behaviour as the reference (``src/attention_crc.jl:4-31``,
``src/attention.jl:133-177``, ``src/attention_bwd.jl:199-275``), with torch-ROCm tensors standing
in for ``ROCArray`` (device memory + streams only -- every FLOP runs in libnnop_hip.so):

    reference (Julia, column-major)          here (torch, row-major, same bytes)
    q,o     (E, QL, QH, B)                   [B, QH, QL, E]
    k,v     (E, KL, KH, B)                   [B, KH, KL, E]
    ms,ls   (QL, QH, B)                      [B, QH, QL]
    pair    (QH, QL, KL, B)                  [B, KL, QL, QH]
    kpad_mask (KL, B) Bool                   [B, KL] torch.bool

The Julia shim that a maintainer of the reference would load instead is
``nnop.jl_amd/julia/NNopHIPExt.jl``; it makes the identical calls with ``ccall``.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math

import torch

from . import _lib
from ._lib import FaDesc

__all__ = [
    "NNopError", "flash_attention", "_flash_attention", "grad_flash_attention",
    "shared_memory", "bwd_workspace_bytes", "fa_fwd_into", "fa_bwd_into",
]


class NNopError(RuntimeError):
    """Counterpart of the ``ErrorException`` raised by the reference's ``error(...)`` calls."""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


_DTYPES = {torch.float32: _lib.NNOP_F32, torch.float16: _lib.NNOP_F16, torch.bfloat16: _lib.NNOP_BF16}


def _jl_shape(t):
    """Shape printed the way the reference prints it (Julia order)."""
    return "(" + ", ".join(str(int(x)) for x in reversed(t.shape)) + ")"


def _raise_status(st, q, k, v):
    """Re-raise a C status with the reference's message (src/attention.jl:141-144)."""
    QE, KE = q.shape[-1], k.shape[-1]
    QH, KH = q.shape[1], k.shape[1]
    if st == _lib.NNOP_ERR_EMB_MISMATCH:
        raise NNopError(f"Embedding dim of Q `{QE}` must be the same as of K `{KE}`.", st)
    if st == _lib.NNOP_ERR_KV_SHAPE:
        raise NNopError(f"Shapes of K `{_jl_shape(k)}` and V `{_jl_shape(v)}` must be the same.", st)
    if st == _lib.NNOP_ERR_EMB_NOT_POW2:
        raise NNopError("Only power-of-2 embedding dims are supported.", st)
    if st == _lib.NNOP_ERR_HEADS:
        raise NNopError(
            f"Number of query heads `{QH}` must be divisible by number of KV heads `{KH}`.", st)
    if st == _lib.NNOP_ERR_EMB_UNSUPPORTED:
        # the reference's counterpart: "Failed to find groupsize ..." (src/attention.jl:204)
        raise NNopError("Failed to find groupsize for Flash Attention that satisfies Shared Memory "
                        f"constraint. ({_lib.strerror(st)})", st)
    raise NNopError(_lib.strerror(st), st)


def _check_inputs(q, k, v, pair, kpad_mask):
    for name, t in (("q", q), ("k", k), ("v", v)):
        if not isinstance(t, torch.Tensor) or t.dim() != 4:
            raise TypeError(f"`{name}` must be a 4-D tensor [B, H, L, E]")
    if not q.is_cuda:
        # the reference's kernels are declared cpu=false (src/attention.jl:1) and its tests
        # refuse to run without a GPU backend (test/runtests.jl:15-17); same here, no fallback.
        raise NNopError("NNop flash attention is GPU-only: tensors must live on a HIP device "
                        "(there is no CPU or PyTorch fallback).")
    if q.dtype not in _DTYPES:
        raise TypeError(f"unsupported element type {q.dtype}; expected float32, float16 or bfloat16")
    others = [k, v] + ([pair] if pair is not None else [])
    for t in others:
        # all of q,k,v,pair share one T in the reference (src/attention.jl:134-137): MethodError
        if t.dtype != q.dtype or t.device != q.device:
            raise TypeError("q, k, v (and pair) must share one dtype and device")
    if pair is not None:
        B, QH, QL, _ = q.shape
        KL = k.shape[2]
        if tuple(pair.shape) != (B, KL, QL, QH):
            raise NNopError(f"pair must have shape [B, KL, QL, QH] = {(B, KL, QL, QH)}, got {tuple(pair.shape)}")
    if kpad_mask is not None:
        if kpad_mask.dtype != torch.bool or kpad_mask.dim() != 2 or kpad_mask.device != q.device:
            raise TypeError("kpad_mask must be a torch.bool matrix [B, KL] on the same device")
        if tuple(kpad_mask.shape) != (q.shape[0], k.shape[2]):
            raise NNopError(f"kpad_mask must have shape [B, KL] = {(q.shape[0], k.shape[2])}")


def _desc(q, k, v, causal):
    B, QH, QL, E = q.shape
    _, KH, KL, KE = k.shape
    return FaDesc(dtype=_DTYPES[q.dtype], emb=E, ql=QL, kl=KL, qh=QH, kh=KH, batch=B,
                  causal=1 if causal else 0, emb_k=KE, emb_v=v.shape[3], kl_v=v.shape[2], kh_v=v.shape[1])


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# Per-call host cost matters for launch-bound shapes (a norm over 1024 x 1024 runs 4 us on the GPU): take the raw stream
# handle without building a torch.cuda.Stream object, and skip the device guard when the tensor's device is current.
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_NO_GUARD = contextlib.nullcontext()


def _stream(t):
    if _raw_stream is not None:
        idx = t.device.index
        return C.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _on_device(t):
    """Context that makes t's device current for allocations and the launch (a no-op when it already is)."""
    idx = t.device.index
    return _NO_GUARD if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(t.device)


def shared_memory(device_id: int = 0) -> int:
    """``NNop._shared_memory(::ROCBackend, device_id)`` (ext/NNopAMDGPUExt.jl:6-9)."""
    out = C.c_uint64(0)
    st = _lib.load().nnop_shared_memory(int(device_id), C.byref(out))
    if st != _lib.NNOP_OK:
        raise NNopError(_lib.strerror(st), st)
    return int(out.value)


def bwd_workspace_bytes(q, k, v, *, causal: bool, pair: bool = False) -> int:
    """``nnop_fa_bwd_workspace_bytes``; ``pair=True``: ``nnop_fa_bwd_workspace_bytes_pair`` -- the scratch with which the
    backward of a call WITH a pair bias runs on 16-byte accesses (the small size still works, through the slow direct path)."""
    lib = _lib.load()
    f = lib.nnop_fa_bwd_workspace_bytes_pair if pair else lib.nnop_fa_bwd_workspace_bytes
    return int(f(C.byref(_desc(q, k, v, causal))))


def fa_fwd_into(o, ms, ls, q, k, v, pair=None, *, causal: bool, kpad_mask=None):
    """Raw ``nnop_fa_fwd`` into caller-owned, preallocated outputs (the C ABI's ownership model:
    the caller allocates everything).  No checks beyond the library's own; contiguous tensors only.
    Used by bench.py so that a timed step is exactly one library call."""
    d = _desc(q, k, v, causal)
    st = _lib.load().nnop_fa_fwd(C.byref(d), _ptr(o), _ptr(ms), _ptr(ls), _ptr(q), _ptr(k), _ptr(v),
                                 _ptr(pair), _ptr(kpad_mask), _stream(q))
    if st != _lib.NNOP_OK:
        _raise_status(st, q, k, v)


def fa_bwd_into(dq, dk, dv, dpair, ws, dO, o, ms, ls, q, k, v, pair=None, *, causal: bool, kpad_mask=None):
    """Raw ``nnop_fa_bwd`` into caller-owned outputs and workspace (see fa_fwd_into)."""
    d = _desc(q, k, v, causal)
    st = _lib.load().nnop_fa_bwd(C.byref(d), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dpair), _ptr(dO), _ptr(o),
                                 _ptr(ms), _ptr(ls), _ptr(q), _ptr(k), _ptr(v), _ptr(pair), _ptr(kpad_mask),
                                 _ptr(ws), C.c_size_t(ws.numel() * ws.element_size()), _stream(q))
    if st != _lib.NNOP_OK:
        _raise_status(st, q, k, v)


def _flash_attention(q, k, v, pair=None, *, causal: bool, kpad_mask=None):
    """``NNop._flash_attention`` (src/attention.jl:133-177): returns ``(o, ms, ls)``.

    Asynchronous on the current torch stream, like the reference's KA launch.
    """
    lib = _lib.load()
    _check_inputs(q, k, v, pair, kpad_mask)
    if k.shape[0] != q.shape[0]:
        raise NNopError(f"Batch of K `{k.shape[0]}` must be the same as of Q `{q.shape[0]}`.")
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    pair = pair.contiguous() if pair is not None else None
    kpad_mask = kpad_mask.contiguous() if kpad_mask is not None else None
    d = _desc(q, k, v, causal)
    B, QH, QL, E = q.shape
    with _on_device(q):
        o = torch.empty_like(q)                                           # similar(q)      :166
        ms = torch.empty((B, QH, QL), dtype=q.dtype, device=q.device)     # KA.allocate     :167
        ls = torch.empty((B, QH, QL), dtype=q.dtype, device=q.device)     # KA.allocate     :168
        st = lib.nnop_fa_fwd(C.byref(d), _ptr(o), _ptr(ms), _ptr(ls), _ptr(q), _ptr(k), _ptr(v),
                             _ptr(pair), _ptr(kpad_mask), _stream(q))
    if st != _lib.NNOP_OK:
        _raise_status(st, q, k, v)
    return o, ms, ls


def grad_flash_attention(dO, o, ms, ls, q, k, v, pair=None, *, causal: bool, kpad_mask=None):
    """``NNop.∇flash_attention`` (src/attention_bwd.jl:199-275): returns ``(dq, dk, dv, dpair|None)``."""
    lib = _lib.load()
    _check_inputs(q, k, v, pair, kpad_mask)
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    dO, o, ms, ls = dO.contiguous(), o.contiguous(), ms.contiguous(), ls.contiguous()
    if dO.dtype != q.dtype or dO.shape != q.shape:
        raise TypeError("cotangent must have the dtype and shape of the output")
    # residuals of the forward: typed like the reference's signature (o::AbstractArray{T,4}, ms/ls::AbstractArray{T,3},
    # src/attention_bwd.jl:199-207) -- the kernels read them as T with these shapes
    B, QH, QL, _ = q.shape
    if o.dtype != q.dtype or o.shape != q.shape or o.device != q.device:
        raise TypeError("`o` must have the dtype, shape and device of `q`")
    for name, t in (("ms", ms), ("ls", ls)):
        if t.dtype != q.dtype or tuple(t.shape) != (B, QH, QL) or t.device != q.device:
            raise TypeError(f"`{name}` must be a [B, QH, QL] = {(B, QH, QL)} tensor of q's dtype on q's device, "
                            f"got {tuple(t.shape)} {t.dtype}")
    if dO.device != q.device:
        raise TypeError("cotangent must live on q's device")
    pair = pair.contiguous() if pair is not None else None
    kpad_mask = kpad_mask.contiguous() if kpad_mask is not None else None
    d = _desc(q, k, v, causal)
    with _on_device(q):
        dq = torch.empty_like(q)
        dk = torch.empty_like(k)
        dv = torch.empty_like(v)
        dpair = torch.empty_like(pair) if pair is not None else None
        nbytes = small = int(lib.nnop_fa_bwd_workspace_bytes(C.byref(d)))
        if nbytes != 0 and pair is not None:
            # staged pair-bias path: two head-major bias-sized scratch matrices on top (the library returns the small size
            # where that path does not exist: plain-HIP embedding dims, too many heads for its LDS block)
            nbytes = max(nbytes, int(lib.nnop_fa_bwd_workspace_bytes_pair(C.byref(d))))
        if nbytes == 0:
            st = lib.nnop_fa_bwd(C.byref(d), *([C.c_void_p(0)] * 13), C.c_void_p(0), 0, C.c_void_p(0))
            _raise_status(st, q, k, v)
        try:
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        except torch.cuda.OutOfMemoryError:
            if nbytes == small:
                raise
            nbytes = small                     # no room for the scratch: the direct (element-wise) pair path needs none
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=q.device)
        st = lib.nnop_fa_bwd(C.byref(d), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(dpair), _ptr(dO), _ptr(o),
                             _ptr(ms), _ptr(ls), _ptr(q), _ptr(k), _ptr(v), _ptr(pair), _ptr(kpad_mask),
                             _ptr(ws), C.c_size_t(nbytes), _stream(q))
    if st != _lib.NNOP_OK:
        _raise_status(st, q, k, v)
    return dq, dk, dv, dpair


class _FlashAttentionFn(torch.autograd.Function):
    """``CRC.rrule(::typeof(_flash_attention), ...)`` (src/attention_crc.jl:16-31): the forward
    closes over (o, ms, ls, q, k, v, pair); the pullback maps Δ to (dq, dk, dv, dpair) and gives
    no tangent for kpad_mask."""

    @staticmethod
    def forward(ctx, q, k, v, pair, kpad_mask, causal):
        o, ms, ls = _flash_attention(q, k, v, pair, causal=causal, kpad_mask=kpad_mask)
        ctx.save_for_backward(o, ms, ls, q, k, v, pair if pair is not None else torch.empty(0),
                              kpad_mask if kpad_mask is not None else torch.empty(0))
        ctx.has_pair, ctx.has_mask, ctx.causal = pair is not None, kpad_mask is not None, bool(causal)
        return o

    @staticmethod
    def backward(ctx, dO):
        o, ms, ls, q, k, v, pair, mask = ctx.saved_tensors
        dq, dk, dv, dpair = grad_flash_attention(
            dO, o, ms, ls, q, k, v, pair if ctx.has_pair else None,
            causal=ctx.causal, kpad_mask=mask if ctx.has_mask else None)
        return dq, dk, dv, dpair, None, None


def flash_attention(q, k, v, pair=None, *, causal: bool, kpad_mask=None):
    """``NNop.flash_attention(q, k, v, pair=nothing; causal, kpad_mask=nothing)``
    (src/attention_crc.jl:4-14).  ``causal`` is a required keyword, as in the reference.
    Returns ``o``; differentiable w.r.t. q, k, v, pair through the rrule above."""
    if torch.is_grad_enabled() and any(
            t is not None and t.requires_grad for t in (q, k, v, pair)):
        return _FlashAttentionFn.apply(q, k, v, pair, kpad_mask, bool(causal))
    return _flash_attention(q, k, v, pair, causal=causal, kpad_mask=kpad_mask)[0]
