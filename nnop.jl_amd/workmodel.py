"""Algorithmic work of one call: the FLOPs / bytes that bench.py and the perf tools divide by the
measured time.  Pure arithmetic on shapes; part of the product so that measurement code never needs the test oracle."""
from __future__ import annotations

__all__ = ["attention_flops", "attention_bytes", "rope_bytes", "softmax_bytes", "norm_bytes"]


def attention_flops(E, QL, KL, QH, B, *, causal: bool, mode: str = "fwd", kv_lens=None):
    """fwd non-causal 4*E*QL*KL*QH*B; causal counts the visible (query, key) pairs; with kv_lens (key padding) the key
    count per batch is sum(len_b); bwd = 2.5x fwd; fwd+bwd = 3.5x fwd."""
    if kv_lens is not None:
        f = 4 * E * QH * QL * int(sum(int(x) for x in kv_lens))
    elif causal:
        n = min(QL, KL)
        pairs = n * (n + 1) // 2 + max(QL - KL, 0) * KL
        f = 4 * E * QH * B * pairs
    else:
        f = 4 * E * QL * KL * QH * B
    return {"fwd": f, "bwd": f * 5 // 2, "fwd+bwd": f * 7 // 2}[mode]


def attention_bytes(E, QL, KL, QH, KH, B, itemsize, *, mode: str = "fwd"):
    """Every tensor touched once: fwd reads q, k, v and writes o, ms, ls; bwd reads q, k, v, o, dO, ms, ls and writes
    dq, dk, dv."""
    nq, nk, st = B * QH * QL * E, B * KH * KL * E, 2 * B * QH * QL
    fwd = itemsize * (nq + 2 * nk + nq + st)
    bwd = itemsize * (nq + 2 * nk + nq + nq + nq + 2 * nk + st)
    return {"fwd": fwd, "bwd": bwd, "fwd+bwd": fwd + bwd}[mode]


def rope_bytes(D, L, QH, KH, B, itemsize):
    """q and k read once and written once, the D/2-wide cos / sin rows (fp32) read once."""
    return 2 * itemsize * B * L * D * (QH + KH) + 2 * 4 * B * L * (D // 2)


def softmax_bytes(N, batch, itemsize, bwd=False):
    """forward reads x and writes y; pullback reads dy, y and writes dx."""
    return (3 if bwd else 2) * N * batch * itemsize


def norm_bytes(emb, n, itemsize, bwd=False):
    """forward reads x, writes y; pullback reads dy, x, writes dx (w, b, statistics are O(emb + n))."""
    return (3 if bwd else 2) * emb * n * itemsize
