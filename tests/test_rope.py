"""CPU tests for the Llama RoPE row: the oracle's two statements of the operator agree
(test/rope_tests.jl:6-19 vs the kernel's pairwise form src/rope/llama_rope.jl:43-61), the pullback is the transpose,
the host mirror of LlamaRotaryEmbedding matches the oracle's, and the C ABI validates descriptors."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle.naive_rope import (llama_rotary_embedding, naive_llama_rope, pairwise_llama_rope, rope_bytes)


def _inputs(seed, B, QH, KH, L, D):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((B, QH, L, D))
    k = rng.standard_normal((B, KH, L, D))
    pos = np.tile(np.arange(L, dtype=np.float32), (B, 1)) + rng.integers(0, 50, size=(B, 1)).astype(np.float32)
    cos, sin = llama_rotary_embedding(D, pos)
    return q, k, cos, sin


@pytest.mark.parametrize("L", [13, 255, 257])
@pytest.mark.parametrize("QH,KH", [(1, 1), (3, 5), (4, 1)])
def test_naive_and_pairwise_forms_agree(L, QH, KH):
    q, k, cos, sin = _inputs(L + QH, 2, QH, KH, L, 16)
    a = naive_llama_rope(q, k, cos, sin)
    b = pairwise_llama_rope(q, k, cos, sin)
    for x, y in zip(a, b):
        np.testing.assert_allclose(x, y, rtol=1e-13, atol=1e-13)


def test_reference_test_case_all_ones():
    """test/rope_tests.jl:29-40: q = k = ones, position ids 0..L-1."""
    D, L, B = 16, 13, 2
    pos = np.tile(np.arange(L, dtype=np.float32), (B, 1))
    cos, sin = llama_rotary_embedding(D, pos)
    q = np.ones((B, 3, L, D)); k = np.ones((B, 1, L, D))
    qo, ko = naive_llama_rope(q, k, cos, sin)
    half = D // 2
    c, s = cos[:, None, :, :half].astype(np.float64), sin[:, None, :, :half].astype(np.float64)
    np.testing.assert_allclose(qo[..., :half], np.broadcast_to(c - s, qo[..., :half].shape), rtol=1e-14)
    np.testing.assert_allclose(ko[..., half:], np.broadcast_to(c + s, ko[..., half:].shape), rtol=1e-14)
    assert np.allclose(qo[:, :, 0], 1.0)           # position 0: identity


def test_pullback_is_transpose_and_inverse():
    q, k, cos, sin = _inputs(3, 2, 3, 2, 37, 32)
    rng = np.random.default_rng(4)
    gq, gk = rng.standard_normal(q.shape), rng.standard_normal(k.shape)
    qo, ko = pairwise_llama_rope(q, k, cos, sin)
    dq, dk = pairwise_llama_rope(gq, gk, cos, sin, sin_sign=-1.0)
    # <rope(q), g> == <q, rope^T(g)>
    np.testing.assert_allclose((qo * gq).sum(), (q * dq).sum(), rtol=1e-12)
    np.testing.assert_allclose((ko * gk).sum(), (k * dk).sum(), rtol=1e-12)
    # a rotation: the transpose is the inverse (up to the fp32 rounding of cos^2 + sin^2) and norms are kept
    q2, k2 = pairwise_llama_rope(qo, ko, cos, sin, sin_sign=-1.0)
    np.testing.assert_allclose(q2, q, atol=1e-6)
    np.testing.assert_allclose(k2, k, atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(qo, axis=-1), np.linalg.norm(q, axis=-1), rtol=1e-6)


def test_embedding_host_mirror_matches_oracle(pkg):
    for dim, base in [(16, 10000), (64, 10000), (128, 500000)]:
        pos = np.tile(np.arange(300, dtype=np.float32), (2, 1))
        cos_o, sin_o = llama_rotary_embedding(dim, pos, base=base)
        emb = pkg.LlamaRotaryEmbedding(dim, base=base)
        cos, sin = emb(torch.tensor(pos))
        assert cos.shape == (2, 300, dim) and cos.dtype == torch.float32
        # fp32 pow / cos of different libms: a few ulp of the argument (<= 300 rad)
        np.testing.assert_allclose(cos.numpy(), cos_o, atol=2e-4)
        np.testing.assert_allclose(sin.numpy(), sin_o, atol=2e-4)
        np.testing.assert_array_equal(cos.numpy()[..., : dim // 2], cos.numpy()[..., dim // 2:])
        np.testing.assert_allclose(emb.inv_freq.numpy()[0], 1.0)


def test_rope_bytes():
    assert rope_bytes(128, 4096, 32, 8, 2, 2) == 2 * 2 * 2 * 4096 * 128 * 40 + 2 * 4 * 2 * 4096 * 64


@pytest.mark.parametrize("kw,status", [
    (dict(dtype=9), "NNOP_ERR_DTYPE"),
    (dict(dtype=0, cs_dtype=1), "NNOP_ERR_DTYPE"),       # cos/sin must be fp32 or T
    (dict(dim=15), "NNOP_ERR_SHAPE"),
    (dict(seq=0), "NNOP_ERR_SHAPE"),
    (dict(kh=-1), "NNOP_ERR_SHAPE"),
    (dict(), "NNOP_ERR_NULL"),
])
def test_rope_descriptor_validation(pkg, kw, status):
    lib = pkg._lib.load()
    base = dict(dtype=2, cs_dtype=0, dim=16, seq=13, qh=3, kh=1, batch=2)
    base.update(kw)
    d = pkg._lib.RopeDesc(**base)
    null = C.c_void_p(0)
    assert lib.nnop_llama_rope(C.byref(d), null, null, null, null, null, null, C.c_float(1.0), null) == \
        getattr(pkg._lib, status)
    assert lib.nnop_llama_rope(None, null, null, null, null, null, null, C.c_float(1.0), null) == pkg._lib.NNOP_ERR_NULL


def test_rope_host_refuses_cpu_tensors(pkg):
    q = torch.ones(1, 1, 4, 16); cos = torch.ones(1, 4, 16)
    with pytest.raises(pkg.NNopError, match="GPU-only"):
        pkg.llama_rope(q, q, cos=cos, sin=cos)


def test_oracle_reproduces_rope_golden_fixtures():
    """The committed fixtures (tests/golden/rope_*.npz, make_golden.py) are what the oracle computes today."""
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "rope_*.npz")))
    assert len(files) >= 3
    for f in files:
        g = np.load(f)
        cos, sin = llama_rotary_embedding(g["q"].shape[-1], g["position_ids"])
        # libm / SIMD cos of another host may differ in the last ulp of fp32
        np.testing.assert_allclose(cos, g["cos"], atol=1e-6); np.testing.assert_allclose(sin, g["sin"], atol=1e-6)
        qo, ko = naive_llama_rope(g["q"], g["k"], g["cos"], g["sin"])
        np.testing.assert_allclose(qo, g["q_out"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(ko, g["k_out"], rtol=1e-6, atol=1e-7)
        dq, dk = pairwise_llama_rope(g["dq_out"], g["dk_out"], g["cos"], g["sin"], sin_sign=-1.0)
        np.testing.assert_allclose(dq, g["dq"], rtol=1e-6, atol=1e-7)
