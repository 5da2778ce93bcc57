"""CPU oracle for the NNop.jl Flash-Attention hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product path (``nnop.jl_amd``) never routes through anything in ``oracle/`` and
fails loudly when the HIP library is missing.

PARITY UNPINNED (with respect to the reference's own outputs): the reference is
Julia + GPU-only kernels (``src/attention.jl:1`` ``cpu=false``); there is no Julia
in the build container or on the GPU box, and the reference's tests hold no golden
vectors (inputs are unseeded ``randn``, ``test/attention_tests.jl:21-23``).  What the
reference's tests DO pin is the property "flash == naive formula" at norm-wise
atol=rtol=1e-3 (``test/attention_tests.jl:42-48``), so the oracle restates that naive
formula (``test/attention_testsetup.jl:21-45``) and the reference's tiled recurrences
(``src/attention.jl:44-130``, ``src/attention_bwd.jl:39-197``).  It is cross-checked in
``tests/test_oracle.py`` against an independent implementation (torch CPU fp64
``softmax(QK^T)V`` + autograd) and frozen by the fixtures in ``tests/golden/``.

Array layout everywhere (row-major / C order, last index fastest):

    q, o, dO, dq : [B, QH, QL, E]      == Julia (E, QL, QH, B)
    k, v, dk, dv : [B, KH, KL, E]      == Julia (E, KL, KH, B)
    ms, ls, delta: [B, QH, QL]         == Julia (QL, QH, B)
    pair, dpair  : [B, KL, QL, QH]     == Julia (QH, QL, KL, B)   (src/attention.jl:62)
    kpad_mask    : [B, KL] bool        == Julia (KL, B)           (src/attention.jl:76)
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "naive_attention",
    "naive_attention_grads",
    "naive_attention_slice",
    "tiled_flash_fwd",
    "tiled_flash_bwd",
    "naive_attention_f32",
    "naive_attention_f32_fwd_bwd",
    "attention_flops",
    "attention_bytes",
]


# --------------------------------------------------------------------------- helpers
def _check_shapes(q, k, v):
    """The four argument checks of src/attention.jl:141-144 (same messages)."""
    B, QH, QL, QE = q.shape
    KB, KH, KL, KE = k.shape
    if QE != KE:
        raise ValueError(f"Embedding dim of Q `{QE}` must be the same as of K `{KE}`.")
    if k.shape != v.shape:
        raise ValueError(f"Shapes of K `{k.shape}` and V `{v.shape}` must be the same.")
    if QE & (QE - 1) != 0 or QE <= 0:
        raise ValueError("Only power-of-2 embedding dims are supported.")
    if QH % KH != 0:
        raise ValueError(
            f"Number of query heads `{QH}` must be divisible by number of KV heads `{KH}`.")
    return B, QH, QL, KH, KL, QE


def _expand_kv(x, n_rep):
    """GQA: query head h uses kv head h // n_rep.

    test/attention_testsetup.jl:23-30 repeats each KV head `num_q_per_kv` times with
    the repeat index fastest (Julia column-major "(num_q_per_kv h)"), which is the
    same mapping as the kernel's `kv_head = cld(q_head, n_q_per_kv)` (src/attention.jl:28).
    """
    if n_rep == 1:
        return x
    return np.repeat(x, n_rep, axis=1)


def _T(x):
    """last two axes swapped (a view)"""
    return np.swapaxes(x, -1, -2)


def _bmm(a, b):
    """batched matrix product over the two leading axes (BLAS-backed np.matmul: the GPU suite evaluates this oracle on ~500 grid
    points, and einsum's generic kernel was 3x slower on the transposed contractions)"""
    return np.matmul(a, b)


def _logits(q, k, pair, causal, kpad_mask, scale):
    """Scaled + biased + masked logits a[b,h,i,j] (i = query, j = key).

    test/attention_testsetup.jl:32-43; mask order in the kernel is
    scale -> +pair -> causal -> pad (src/attention.jl:55-79), equivalent because
    -inf + finite = -inf.
    """
    B, QH, QL, E = q.shape
    KL = k.shape[2]
    a = _bmm(q, _T(k)) * scale
    if pair is not None:
        # pair[b, j, i, h]  ->  [b, h, i, j]
        a = a + np.transpose(pair, (0, 3, 2, 1))
    if causal:
        # keep iff key index <= query index (src/attention.jl:70), top-left aligned
        keep = np.arange(KL)[None, :] <= np.arange(QL)[:, None]
        a = np.where(keep[None, None], a, -np.inf)
    if kpad_mask is not None:
        a = np.where(np.asarray(kpad_mask, dtype=bool)[:, None, None, :], a, -np.inf)
    return a


def _softmax_lastdim(a):
    """test/attention_testsetup.jl:10-14 (softmax over the key axis)."""
    with np.errstate(invalid="ignore"):
        mx = np.max(a, axis=-1, keepdims=True)
        t = np.exp(a - mx)
        return t / np.sum(t, axis=-1, keepdims=True), mx[..., 0], np.sum(t, axis=-1)


# --------------------------------------------------------------------------- naive fwd
def naive_attention(q, k, v, pair=None, *, causal: bool, kpad_mask=None,
                    dtype=np.float64, return_stats: bool = False):
    """Naive attention, restating test/attention_testsetup.jl:21-45 in `dtype`.

    Returns o [B,QH,QL,E]; with return_stats also (ms, ls) with the meaning of
    src/attention.jl:128-129: ms = row max of the scaled+biased+masked logits,
    ls = sum_j exp(a_ij - ms_i).
    """
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    if pair is not None:
        pair = np.asarray(pair, dtype=dtype)
    B, QH, QL, KH, KL, E = _check_shapes(q, k, v)
    n_rep = QH // KH
    ke, ve = _expand_kv(k, n_rep), _expand_kv(v, n_rep)
    scale = dtype(1.0) / np.sqrt(dtype(E))
    a = _logits(q, ke, pair, causal, kpad_mask, scale)
    p, ms, ls = _softmax_lastdim(a)
    o = _bmm(p, ve)
    if return_stats:
        return o, ms, ls
    return o


# --------------------------------------------------------------------------- naive bwd
def naive_attention_grads(q, k, v, dO, pair=None, *, causal: bool, kpad_mask=None,
                          dtype=np.float64):
    """Analytic gradients of naive_attention w.r.t. q, k, v, pair for cotangent dO.

    Same formulas the reference's backward evaluates tile by tile
    (src/attention_bwd.jl:86-156): P = softmax(a); dV = P^T dO; dP = dO V^T;
    dS = P o (dP - rowsum(dO o O)); dpair = dS (:123-132, "dS / scale" undoes the
    scale folded into dS at :118); dQ = scale dS K; dK = scale dS^T Q; GQA sums the
    q-heads of a group into their shared kv head (:99-103, :138-142).
    Returns (dq, dk, dv, dpair or None).
    """
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    dO = np.asarray(dO, dtype=dtype)
    if pair is not None:
        pair = np.asarray(pair, dtype=dtype)
    B, QH, QL, KH, KL, E = _check_shapes(q, k, v)
    n_rep = QH // KH
    ke, ve = _expand_kv(k, n_rep), _expand_kv(v, n_rep)
    scale = dtype(1.0) / np.sqrt(dtype(E))
    a = _logits(q, ke, pair, causal, kpad_mask, scale)
    p, _, _ = _softmax_lastdim(a)
    o = _bmm(p, ve)
    dv_e = _bmm(_T(p), dO)
    dp = _bmm(dO, _T(ve))
    delta = np.sum(dO * o, axis=-1, keepdims=True)
    ds = p * (dp - delta)
    dq = _bmm(ds, ke) * scale
    dk_e = _bmm(_T(ds), q) * scale
    dk = dk_e.reshape(B, KH, n_rep, KL, E).sum(axis=2)
    dv = dv_e.reshape(B, KH, n_rep, KL, E).sum(axis=2)
    dpair = None
    if pair is not None:
        dpair = np.ascontiguousarray(np.transpose(ds, (0, 3, 2, 1)))  # [B,KL,QL,QH]
    return dq, dk, dv, dpair


# --------------------------------------------------------------------------- one (batch, kv-head) slice
def naive_attention_slice(q, k, v, dO=None, *, causal: bool, kpad_mask=None, chunk: int = 1024):
    """The naive formula (test/attention_testsetup.jl:21-45) and its analytic gradients
    (src/attention_bwd.jl:86-156, as in naive_attention_grads) for ONE (batch, kv-head) slice,
    evaluated in fp64 over query-row chunks so that the L x L score matrix of a BASELINE-sized
    problem (L = 8192 .. 16384) never exists at once.  Same math, same order of the masks; every
    (batch, kv-head) slice is independent (src/attention.jl:27-28,33), so a slice of the full-size
    launch can be checked without evaluating the rest.

        q, dO : [G, QL, E]   the G = QH/KH query heads that share this kv head
        k, v  : [KL, E]      kpad_mask : [KL] bool or None

    Returns dict(o [G,QL,E], ms [G,QL], ls [G,QL]) and, when dO is given, dq [G,QL,E], dk, dv [KL,E]
    (dk, dv summed over the G query heads, src/attention_bwd.jl:99-103,138-142).
    """
    q = np.asarray(q, np.float64)
    k = np.asarray(k, np.float64)
    v = np.asarray(v, np.float64)
    G, QL, E = q.shape
    KL = k.shape[0]
    scale = 1.0 / np.sqrt(np.float64(E))
    out = dict(o=np.empty_like(q), ms=np.empty((G, QL)), ls=np.empty((G, QL)))
    if dO is not None:
        dO = np.asarray(dO, np.float64)
        out.update(dq=np.empty_like(q), dk=np.zeros_like(k), dv=np.zeros_like(v))
    kidx = np.arange(KL)
    valid = None if kpad_mask is None else np.asarray(kpad_mask, bool)
    for g in range(G):
        for q0 in range(0, QL, chunk):
            q1 = min(q0 + chunk, QL)
            kl = min(KL, q1) if causal else KL           # keys beyond the chunk's last row are masked anyway
            a = (q[g, q0:q1] @ k[:kl].T) * scale
            if causal:
                a = np.where(kidx[None, :kl] <= np.arange(q0, q1)[:, None], a, -np.inf)
            if valid is not None:
                a = np.where(valid[None, :kl], a, -np.inf)
            with np.errstate(invalid="ignore"):
                mx = a.max(axis=1, keepdims=True)
                t = np.exp(a - mx)
                l = t.sum(axis=1, keepdims=True)
                p = t / l
            o = p @ v[:kl]
            out["o"][g, q0:q1] = o
            out["ms"][g, q0:q1] = mx[:, 0]
            out["ls"][g, q0:q1] = l[:, 0]
            if dO is not None:
                d = dO[g, q0:q1]
                out["dv"][:kl] += p.T @ d
                dp = d @ v[:kl].T
                ds = p * (dp - np.sum(d * o, axis=1, keepdims=True))
                out["dq"][g, q0:q1] = (ds @ k[:kl]) * scale
                out["dk"][:kl] += (ds.T @ q[g, q0:q1]) * scale
    return out


# --------------------------------------------------------------------------- tiled fwd
def tiled_flash_fwd(q, k, v, pair=None, *, causal: bool, kpad_mask=None, gsz: int = 64,
                    dtype=np.float64):
    """Tile-by-tile restatement of `_flash_attention_fwd!` (src/attention.jl:44-130).

    Keeps the reference's recurrences literally: per kv tile m_ij / l_ij (:82-94),
    m_new, alpha, beta, l_new (:97-100), P *= beta/l_new and O *= l*alpha/l_new (:102-110),
    O += P V (:115) -- i.e. O is kept normalised after every tile (FA-1 style).
    One deliberate deviation, documented in SURVEY.md section 7 "hard parts" (iii): a kv
    tile that is fully masked for a row is skipped for that row instead of producing
    exp(-inf - -inf) = NaN, which is what the naive formula gives (finite result).
    Causal iterates kv tiles up to the diagonal AND up to KL (the reference's
    `end_iter = gidx[1]`, :47, assumes QL == KL).
    Returns (o, ms, ls).
    """
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    B, QH, QL, KH, KL, E = _check_shapes(q, k, v)
    n_rep = QH // KH
    scale = dtype(1.0) / np.sqrt(dtype(E))
    o = np.zeros_like(q)
    ms = np.full((B, QH, QL), -np.inf, dtype=dtype)
    ls = np.zeros((B, QH, QL), dtype=dtype)
    for b in range(B):
        for h in range(QH):
            kh = h // n_rep
            for q0 in range(0, QL, gsz):
                q1 = min(q0 + gsz, QL)
                qt = q[b, h, q0:q1]
                m_i = np.full((q1 - q0,), -np.inf, dtype=dtype)
                l_i = np.zeros((q1 - q0,), dtype=dtype)
                o_t = np.zeros((q1 - q0, E), dtype=dtype)
                for k0 in range(0, KL, gsz):
                    if causal and k0 > q1 - 1:
                        break
                    k1 = min(k0 + gsz, KL)
                    s = (qt @ k[b, kh, k0:k1].T) * scale
                    if pair is not None:
                        s = s + np.asarray(pair[b, k0:k1, q0:q1, h], dtype=dtype).T
                    if causal:
                        keep = np.arange(k0, k1)[None, :] <= np.arange(q0, q1)[:, None]
                        s = np.where(keep, s, -np.inf)
                    if kpad_mask is not None:
                        s = np.where(np.asarray(kpad_mask[b, k0:k1], bool)[None, :], s, -np.inf)
                    m_ij = s.max(axis=1)
                    live = np.isfinite(m_ij)                   # deviation (iii): skip dead rows
                    m_safe = np.where(live, m_ij, 0.0)
                    p = np.exp(s - m_safe[:, None])
                    p = np.where(live[:, None], p, 0.0)
                    l_ij = p.sum(axis=1)
                    m_new = np.maximum(m_i, m_ij)
                    m_new_safe = np.where(np.isfinite(m_new), m_new, 0.0)
                    alpha = np.where(np.isfinite(m_i), np.exp(m_i - m_new_safe), 0.0)
                    beta = np.where(live, np.exp(m_safe - m_new_safe), 0.0)
                    l_new = alpha * l_i + beta * l_ij
                    with np.errstate(invalid="ignore", divide="ignore"):
                        p_scale = np.where(l_new > 0, beta / l_new, 0.0)
                        o_scale = np.where(l_new > 0, l_i / l_new * alpha, 0.0)
                    o_t = o_t * o_scale[:, None] + (p * p_scale[:, None]) @ v[b, kh, k0:k1]
                    m_i, l_i = m_new, l_new
                dead = l_i == 0
                if np.any(dead):      # rows with no visible key: naive gives 0/0 = NaN
                    o_t = np.where(dead[:, None], np.nan, o_t)
                o[b, h, q0:q1] = o_t
                ms[b, h, q0:q1] = m_i
                ls[b, h, q0:q1] = l_i
    return o, ms, ls


# --------------------------------------------------------------------------- tiled bwd
def tiled_flash_bwd(dO, o, ms, ls, q, k, v, pair=None, *, causal: bool, kpad_mask=None,
                    gsz: int = 64, dtype=np.float64):
    """Tile-by-tile restatement of the reference backward.

    Preprocess (src/attention_bwd.jl:163-197): D_scaled = dO / ls ; delta = sum_e D_scaled*o.
    Main (src/attention_bwd.jl:39-160), per kv tile n and q tile m (causal: m >= n):
      P~ = exp(S - ms)                  (:86-90, un-normalised, 1/ls lives in D_scaled)
      dV += P~^T D_scaled               (:94-105)
      dS  = P~ o (D_scaled V^T - delta) * scale        (:108-120)
      dpair = dS / scale                (:123-132)
      dK += dS^T Q ; dQ += dS K         (:134-156)
    Deviation: Q and K are NOT rounded to Float16 (the reference does, :19-20,44,54).
    Returns (dq, dk, dv, dpair or None).
    """
    q = np.asarray(q, dtype=dtype)
    k = np.asarray(k, dtype=dtype)
    v = np.asarray(v, dtype=dtype)
    dO = np.asarray(dO, dtype=dtype)
    o = np.asarray(o, dtype=dtype)
    ms = np.asarray(ms, dtype=dtype)
    ls = np.asarray(ls, dtype=dtype)
    B, QH, QL, KH, KL, E = _check_shapes(q, k, v)
    n_rep = QH // KH
    scale = dtype(1.0) / np.sqrt(dtype(E))
    d_scaled = dO / ls[..., None]
    delta = np.sum(d_scaled * o, axis=-1)
    dq = np.zeros_like(q)
    dk = np.zeros_like(k)
    dv = np.zeros_like(v)
    dpair = None if pair is None else np.zeros((B, KL, QL, QH), dtype=dtype)
    for b in range(B):
        for h in range(QH):
            kh = h // n_rep
            for k0 in range(0, KL, gsz):
                k1 = min(k0 + gsz, KL)
                kt, vt = k[b, kh, k0:k1], v[b, kh, k0:k1]
                q_start = (k0 // gsz) * gsz if causal else 0
                for q0 in range(q_start, QL, gsz):
                    q1 = min(q0 + gsz, QL)
                    qt = q[b, h, q0:q1]
                    s = (qt @ kt.T) * scale
                    if pair is not None:
                        s = s + np.asarray(pair[b, k0:k1, q0:q1, h], dtype=dtype).T
                    if causal:
                        keep = np.arange(k0, k1)[None, :] <= np.arange(q0, q1)[:, None]
                        s = np.where(keep, s, -np.inf)
                    if kpad_mask is not None:
                        s = np.where(np.asarray(kpad_mask[b, k0:k1], bool)[None, :], s, -np.inf)
                    with np.errstate(invalid="ignore"):
                        pt = np.exp(s - ms[b, h, q0:q1, None])
                    pt = np.where(np.isfinite(s), pt, 0.0)
                    dsc = d_scaled[b, h, q0:q1]
                    dv[b, kh, k0:k1] += pt.T @ dsc
                    ds = pt * (dsc @ vt.T - delta[b, h, q0:q1, None]) * scale
                    if dpair is not None:
                        dpair[b, k0:k1, q0:q1, h] = (ds / scale).T
                    dk[b, kh, k0:k1] += ds.T @ qt
                    dq[b, h, q0:q1] += ds @ kt
    return dq, dk, dv, dpair


# --------------------------------------------------------------------------- cpu baseline
def naive_attention_f32(q, k, v, *, causal: bool = False):
    """fp32 naive attention of benchmarks/main.jl:26-43 (Float32 scale, batched GEMMs,
    softmax over keys), restated with numpy BLAS so it uses the host's cores the way
    NNlib's batched_mul would.  Non-GQA (the benchmark twin has no GQA).  This is the
    `cpu_baseline` that bench.py times; materialises the [B,H,QL,KL] score tensor."""
    q = np.asarray(q, dtype=np.float32)
    k = np.asarray(k, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    E = q.shape[-1]
    a = np.matmul(q, np.swapaxes(k, -1, -2))
    a *= np.float32(1.0 / np.sqrt(E))
    if causal:
        QL, KL = q.shape[2], k.shape[2]
        keep = np.arange(KL)[None, :] <= np.arange(QL)[:, None]
        a = np.where(keep[None, None], a, np.float32(-np.inf))
    a -= a.max(axis=-1, keepdims=True)
    np.exp(a, out=a)
    a /= a.sum(axis=-1, keepdims=True)
    return np.matmul(a, v)


def naive_attention_f32_fwd_bwd(q, k, v, dO, *, causal: bool = False):
    """Forward + backward of naive_attention_f32 with explicit fp32 GEMMs (what Zygote
    generates for benchmarks/main.jl:365-373): returns (o, dq, dk, dv)."""
    q = np.asarray(q, dtype=np.float32)
    k = np.asarray(k, dtype=np.float32)
    v = np.asarray(v, dtype=np.float32)
    dO = np.asarray(dO, dtype=np.float32)
    E = q.shape[-1]
    scale = np.float32(1.0 / np.sqrt(E))
    a = np.matmul(q, np.swapaxes(k, -1, -2))
    a *= scale
    if causal:
        QL, KL = q.shape[2], k.shape[2]
        keep = np.arange(KL)[None, :] <= np.arange(QL)[:, None]
        a = np.where(keep[None, None], a, np.float32(-np.inf))
    a -= a.max(axis=-1, keepdims=True)
    np.exp(a, out=a)
    a /= a.sum(axis=-1, keepdims=True)
    o = np.matmul(a, v)
    dv = np.matmul(np.swapaxes(a, -1, -2), dO)
    dp = np.matmul(dO, np.swapaxes(v, -1, -2))
    delta = np.sum(dO * o, axis=-1, keepdims=True)
    dp -= delta
    dp *= a                      # dS
    dp *= scale
    dq = np.matmul(dp, k)
    dk = np.matmul(np.swapaxes(dp, -1, -2), q)
    return o, dq, dk, dv


# --------------------------------------------------------------------------- work model
def attention_flops(E, QL, KL, QH, B, *, causal: bool, mode: str = "fwd", kv_lens=None):
    """Algorithmic FLOPs: fwd non-causal 4*E*QL*KL*QH*B, causal
    4*E*QH*B*L(L+1)/2 ; bwd = 2.5x fwd ; fwd+bwd = 3.5x fwd.  With kv_lens (key padding)
    the key count per batch is sum(len_b)."""
    if kv_lens is not None:
        f = 4 * E * QH * QL * int(np.sum(kv_lens))
    elif causal:
        n = min(QL, KL)
        pairs = n * (n + 1) // 2 + max(QL - KL, 0) * KL
        f = 4 * E * QH * B * pairs
    else:
        f = 4 * E * QL * KL * QH * B
    return {"fwd": f, "bwd": f * 5 // 2, "fwd+bwd": f * 7 // 2}[mode]


def attention_bytes(E, QL, KL, QH, KH, B, itemsize, *, mode: str = "fwd"):
    """Algorithmic bytes: every tensor touched once."""
    nq, nk, st = B * QH * QL * E, B * KH * KL * E, 2 * B * QH * QL
    fwd = itemsize * (nq + 2 * nk + nq + st)
    bwd = itemsize * (nq + 2 * nk + nq + nq + nq + 2 * nk + st)
    return {"fwd": fwd, "bwd": bwd, "fwd+bwd": fwd + bwd}[mode]
