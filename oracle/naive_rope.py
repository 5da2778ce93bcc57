"""CPU oracle for NNop.jl's Llama RoPE.  TEST INFRASTRUCTURE ONLY (see
oracle/naive_attention.py for the rules: only tests/, smoke() and bench baselines may import this).

PARITY UNPINNED w.r.t. the reference's outputs for the same reason as the attention oracle (Julia, GPU-only kernel
`llama_rope!` declared cpu=false, src/rope/llama_rope.jl:24; no golden vectors, test/rope_tests.jl uses all-ones inputs).
The oracle restates the reference TEST's naive formula (test/rope_tests.jl:6-19) and the embedding
(src/rope/llama_rope.jl:7-22); tests/test_rope.py cross-checks it against the kernel's pairwise form
(src/rope/llama_rope.jl:43-61).

Layout (row-major, last index fastest):  q [B, QH, L, D], k [B, KH, L, D]  == Julia (D, L, H, B);
cos, sin [B, L, D] == Julia (D, L, B); position_ids [B, L] == Julia (L, B).
"""
from __future__ import annotations

import numpy as np

__all__ = ["llama_rotary_embedding", "rotate_half", "naive_llama_rope", "pairwise_llama_rope", "rope_bytes"]


def llama_rotary_embedding(dim: int, position_ids, base: int = 10000, dtype=np.float32):
    """LlamaRotaryEmbedding(dim; base)(position_ids) -> (cos, sin), src/rope/llama_rope.jl:7-22.
    inv_freq = 1 / base^(i/dim), i = 0, 2, ..; freqs = vcat(inv_freq * pos, inv_freq * pos)."""
    ids = np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)
    inv_freq = (np.float32(1.0) / (np.float32(base) ** ids)).astype(np.float32)
    pos = np.asarray(position_ids, dtype=np.float32)[..., None]                 # [B, L, 1]
    freqs = (pos * inv_freq).astype(np.float32)                                 # [B, L, dim/2]
    freqs = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(freqs.astype(dtype)), np.sin(freqs.astype(dtype))


def rotate_half(x):
    """test/rope_tests.jl:6-11: vcat(-x2, x1) along the head dim."""
    half = x.shape[-1] // 2
    return np.concatenate([-x[..., half:], x[..., :half]], axis=-1)


def naive_llama_rope(q, k, cos, sin, dtype=np.float64):
    """test/rope_tests.jl:13-19: q*cos + rotate_half(q)*sin (same for k), cos/sin broadcast over heads."""
    q, k = np.asarray(q, dtype), np.asarray(k, dtype)
    c, s = np.asarray(cos, dtype)[:, None], np.asarray(sin, dtype)[:, None]     # [B, 1, L, D]
    return q * c + rotate_half(q) * s, k * c + rotate_half(k) * s


def pairwise_llama_rope(q, k, cos, sin, sin_sign=1.0, dtype=np.float64):
    """The kernel's own form (src/rope/llama_rope.jl:43-61): only cos[i], sin[i] with i < D/2 are read;
    out[i] = x1*c - x2*s ; out[i + D/2] = x2*c + x1*s ; the pullback is the same with sin_sign = -1 (:92)."""
    outs = []
    for x in (q, k):
        x = np.asarray(x, dtype)
        half = x.shape[-1] // 2
        c = np.asarray(cos, dtype)[:, None, :, :half]
        s = np.asarray(sin, dtype)[:, None, :, :half] * sin_sign
        x1, x2 = x[..., :half], x[..., half:]
        outs.append(np.concatenate([x1 * c - x2 * s, x2 * c + x1 * s], axis=-1))
    return outs[0], outs[1]


def rope_bytes(D, L, QH, KH, B, itemsize):
    """Algorithmic bytes: q and k read once and written once, cos/sin half-rows read once (fp32)."""
    return 2 * itemsize * B * L * D * (QH + KH) + 2 * 4 * B * L * (D // 2)
