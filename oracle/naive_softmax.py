"""CPU oracle for NNop.online_softmax.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED w.r.t. the reference's outputs (GPU-only Julia kernel, `cpu=false`, src/softmax.jl:19; no golden
vectors; test/softmax_tests.jl compares against the naive formula restated here).

Layout: x [batch, N] row-major == Julia (N, batch); softmax along the last axis (Julia dims = 1).
"""
from __future__ import annotations

import numpy as np

__all__ = ["naive_softmax", "online_softmax_md", "naive_softmax_grad", "softmax_bytes"]


def naive_softmax(x, dtype=np.float64):
    """test/softmax_tests.jl:6-10."""
    x = np.asarray(x, dtype)
    mx = x.max(axis=-1, keepdims=True)
    tmp = np.exp(x - mx)
    return tmp / tmp.sum(axis=-1, keepdims=True)


def _md_reduce(a, b):
    """src/softmax.jl:6-16 on (m, d) pairs, with the NaN guard for (-Inf) - (-Inf)."""
    big, small = (a, b) if a[0] > b[0] else (b, a)
    diff = small[0] - big[0]
    if np.isnan(diff):
        diff = -np.inf
    return (big[0], big[1] + small[1] * np.exp(diff))


def online_softmax_md(x, gsz=256):
    """The reference kernel's own evaluation order (src/softmax.jl:33-57): each of `gsz` threads folds its strided
    elements into an (m, d) pair with md_reduce, the pairs are tree-reduced (src/groupreduce.jl:27-37), then
    y = exp(x - m) / d.  Pure-Python loops: small inputs only."""
    x = np.asarray(x, np.float64)
    out = np.empty_like(x)
    for b in range(x.shape[0]):
        parts = []
        for t in range(gsz):
            md = (-np.inf, 0.0)
            for e in range(t, x.shape[1], gsz):
                md = _md_reduce(md, (x[b, e], 1.0))
            parts.append(md)
        s = gsz // 2
        while s > 0:
            for t in range(s):
                parts[t] = _md_reduce(parts[t], parts[t + s])
            s >>= 1
        m, d = parts[0]
        out[b] = np.exp(x[b] - m) / d
    return out


def naive_softmax_grad(dy, y, dtype=np.float64):
    """∇online_softmax, src/softmax.jl:70-80: dx = dy*y - y*sum(dy*y)."""
    dy, y = np.asarray(dy, dtype), np.asarray(y, dtype)
    tmp = dy * y
    return tmp - y * tmp.sum(axis=-1, keepdims=True)


def softmax_bytes(N, batch, itemsize, bwd=False):
    """Algorithmic bytes: forward reads x and writes y once; backward reads dy, y and writes dx."""
    return (3 if bwd else 2) * N * batch * itemsize
