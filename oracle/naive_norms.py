"""CPU oracle for NNop's RMSNorm and LayerNorm.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED w.r.t. the reference's outputs (GPU-only Julia kernels, `cpu=false`, src/rms_norm.jl:3,
src/layer_norm.jl:8; no golden vectors; the reference's tests compare against the naive formulas restated here).

Layout: x [n, emb] row-major == Julia (emb, n); w, b [emb].
"""
from __future__ import annotations

import numpy as np

__all__ = ["naive_rms_norm", "naive_rms_norm_grads", "naive_layer_norm", "naive_layer_norm_grads", "norm_bytes"]


def naive_rms_norm(x, w, offset=0.0, eps=1e-6, dtype=np.float64):
    """test/rmsnorm_tests.jl:7-9.  Returns (y, rstd) -- rstd is the `rms` cache of src/rms_norm.jl:27."""
    x, w = np.asarray(x, dtype), np.asarray(w, dtype)
    rstd = 1.0 / np.sqrt((x * x).mean(axis=-1, keepdims=True) + dtype(eps))
    return (w + dtype(offset)) * x * rstd, rstd[..., 0]


def naive_rms_norm_grads(dy, x, w, offset=0.0, eps=1e-6, dtype=np.float64):
    """The pullback the reference kernel evaluates (src/rms_norm.jl:40-42, 72-101):
    m = dy*(w+offset); dd = sum(m*x); dx = rstd*m - rstd^3*dd/N*x; dw = sum_rows dy*x*rstd."""
    dy, x, w = np.asarray(dy, dtype), np.asarray(x, dtype), np.asarray(w, dtype)
    N = x.shape[-1]
    rstd = 1.0 / np.sqrt((x * x).mean(axis=-1, keepdims=True) + dtype(eps))
    m = dy * (w + dtype(offset))
    dd = (m * x).sum(axis=-1, keepdims=True)
    dx = rstd * m - rstd ** 3 * dd / N * x
    dw = (dy * x * rstd).sum(axis=0)
    return dx, dw


def naive_layer_norm(x, w, b, eps=1e-6, dtype=np.float64):
    """test/layernorm_tests.jl:7-11 (var with corrected=false).  Returns (y, mu, rstd)."""
    x, w, b = np.asarray(x, dtype), np.asarray(w, dtype), np.asarray(b, dtype)
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + dtype(eps))
    return (x - mu) * rstd * w + b, mu[..., 0], rstd[..., 0]


def naive_layer_norm_grads(dy, x, w, eps=1e-6, dtype=np.float64):
    """src/layer_norm.jl:97-133: xn = (x-mu)*rstd; wdy = w*dy; c1 = mean(wdy*xn); c2 = mean(wdy);
    dx = (wdy - (xn*c1 + c2))*rstd; dw = sum_rows dy*xn; db = sum_rows dy."""
    dy, x, w = np.asarray(dy, dtype), np.asarray(x, dtype), np.asarray(w, dtype)
    mu = x.mean(axis=-1, keepdims=True)
    rstd = 1.0 / np.sqrt(((x - mu) ** 2).mean(axis=-1, keepdims=True) + dtype(eps))
    xn = (x - mu) * rstd
    wdy = w * dy
    c1 = (wdy * xn).mean(axis=-1, keepdims=True)
    c2 = wdy.mean(axis=-1, keepdims=True)
    dx = (wdy - (xn * c1 + c2)) * rstd
    return dx, (dy * xn).sum(axis=0), dy.sum(axis=0)


def norm_bytes(emb, n, itemsize, bwd=False):
    """Algorithmic bytes: forward reads x, writes y; pullback reads dy, x, writes dx (w, b, statistics: O(emb + n))."""
    return (3 if bwd else 2) * emb * n * itemsize
