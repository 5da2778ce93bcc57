"""GPU parity tests for Llama RoPE through the C ABI, against oracle/naive_rope.py (fp64 on the rounded inputs).

Grid of the reference's own test (test/rope_tests.jl:21-56): dim 16, L in {13, 255, 256, 257, 1024, 1025},
QH, KH in {1, 3, 4, 5}, batch 2, atol = rtol = 1e-6 for Float32 -- widened with 16-bit element types, large head dims,
odd half-dims (generic path), cos/sin in T, in-place use, non-trivial inputs (the reference uses all-ones)."""
import numpy as np
import pytest
import torch

from oracle.naive_rope import llama_rotary_embedding, naive_llama_rope, pairwise_llama_rope
from util import TORCH_DT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# fp32: the reference's own tolerance (test/rope_tests.jl:38-39).  16-bit: one rounding of an O(|x|) result.
TOL = {"f32": dict(rtol=1e-6, atol=1e-6), "f16": dict(rtol=1e-3, atol=1e-3), "bf16": dict(rtol=8e-3, atol=8e-3)}


def _mk(seed, B, QH, KH, L, D, dt, ones=False, cs_in_t=False):
    rng = np.random.default_rng(seed)
    tdt = TORCH_DT[dt]
    gen = (lambda *s: np.ones(s, np.float32)) if ones else (lambda *s: rng.standard_normal(s).astype(np.float32))
    q = torch.tensor(gen(B, QH, L, D)).to(tdt).to(DEV)
    k = torch.tensor(gen(B, KH, L, D)).to(tdt).to(DEV)
    pos = np.tile(np.arange(L, dtype=np.float32), (B, 1))
    cos, sin = llama_rotary_embedding(D, pos)
    cos, sin = torch.tensor(cos).to(DEV), torch.tensor(sin).to(DEV)
    if cs_in_t:
        cos, sin = cos.to(tdt), sin.to(tdt)
    return q, k, cos, sin


def _np(t):
    return t.detach().to(torch.float64).cpu().numpy()


def _check(got, ref, dt):
    np.testing.assert_allclose(_np(got), ref, **TOL[dt])


@pytest.mark.parametrize("L", [13, 255, 256, 257, 1024, 1025])
@pytest.mark.parametrize("QH", [1, 3, 4, 5])
@pytest.mark.parametrize("KH", [1, 3, 4, 5])
def test_reference_grid_f32(pkg, L, QH, KH):
    """test/rope_tests.jl:21-56 (forward and pullback), all-ones inputs like the reference and random ones."""
    for ones in (True, False):
        q, k, cos, sin = _mk(L * 31 + QH * 7 + KH, 2, QH, KH, L, 16, "f32", ones=ones)
        q.requires_grad_(True); k.requires_grad_(True)
        qo, ko = pkg.llama_rope(q, k, cos=cos, sin=sin)
        rq, rk = naive_llama_rope(_np(q), _np(k), _np(cos), _np(sin))
        _check(qo, rq, "f32"); _check(ko, rk, "f32")
        # test/rope_tests.jl:43-55: gradient of sum(q') + sum(k')
        (qo.sum() + ko.sum()).backward()
        gq, gk = pairwise_llama_rope(np.ones(q.shape), np.ones(k.shape), _np(cos), _np(sin), sin_sign=-1.0)
        _check(q.grad, gq, "f32"); _check(k.grad, gk, "f32")


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("D", [2, 6, 16, 24, 64, 128, 256])
@pytest.mark.parametrize("cs_in_t", [False, True])
def test_dims_and_dtypes(pkg, dt, D, cs_in_t):
    q, k, cos, sin = _mk(D, 2, 4, 2, 193, D, dt, cs_in_t=cs_in_t)
    qo, ko = pkg.llama_rope(q, k, cos=cos, sin=sin)
    assert qo.dtype == q.dtype and qo.shape == q.shape and qo.data_ptr() != q.data_ptr()
    rq, rk = naive_llama_rope(_np(q), _np(k), _np(cos), _np(sin))
    _check(qo, rq, dt); _check(ko, rk, dt)
    g = torch.randn_like(qo), torch.randn_like(ko)
    dq, dk = pkg.grad_llama_rope(g, cos, sin)
    gq, gk = pairwise_llama_rope(_np(g[0]), _np(g[1]), _np(cos), _np(sin), sin_sign=-1.0)
    _check(dq, gq, dt); _check(dk, gk, dt)


def test_f32_is_exact_to_one_ulp_of_fma(pkg):
    """fp32 path: the only freedom is fma contraction of a*c - b*s; compare with fp64 at 2 ulp of the operands."""
    q, k, cos, sin = _mk(5, 2, 8, 2, 777, 128, "f32")
    qo, ko = pkg.llama_rope(q, k, cos=cos, sin=sin)
    rq, _ = naive_llama_rope(_np(q), _np(k), _np(cos), _np(sin))
    half = 64
    mag = np.abs(_np(q)[..., :half]) + np.abs(_np(q)[..., half:])
    err = np.abs(_np(qo) - rq)
    assert (err[..., :half] <= 2 * np.finfo(np.float32).eps * mag + 1e-30).all()
    assert (err[..., half:] <= 2 * np.finfo(np.float32).eps * mag + 1e-30).all()


def test_in_place_and_inputs_untouched(pkg):
    q, k, cos, sin = _mk(9, 2, 3, 1, 300, 64, "bf16")
    q0, k0 = q.clone(), k.clone()
    qo, ko = pkg.llama_rope(q, k, cos=cos, sin=sin)
    assert torch.equal(q, q0) and torch.equal(k, k0)                # the reference rotates copies (:75-76)
    pkg.llama_rope_into(q, k, q, k, cos, sin)                       # in place is allowed by the ABI
    assert torch.equal(q, qo) and torch.equal(k, ko)


def test_round_trip_and_norms_full_size(pkg):
    """Size-independent properties at a Llama-scale shape: ∇rope(rope(x)) == x, row norms preserved, position 0 is
    the identity, heads of one (position, batch) share the rotation."""
    B, QH, KH, L, D = 2, 32, 8, 4096, 128
    g = torch.Generator(device=DEV).manual_seed(1)
    q = torch.randn(B, QH, L, D, device=DEV, generator=g)
    k = torch.randn(B, KH, L, D, device=DEV, generator=g)
    cos, sin = pkg.LlamaRotaryEmbedding(D)(torch.arange(L, device=DEV, dtype=torch.float32).expand(B, L))
    qo, ko = pkg.llama_rope(q, k, cos=cos, sin=sin)
    q2, k2 = pkg.grad_llama_rope((qo, ko), cos, sin)
    assert torch.allclose(q2, q, atol=2e-6, rtol=2e-6) and torch.allclose(k2, k, atol=2e-6, rtol=2e-6)
    assert torch.allclose(qo.norm(dim=-1), q.norm(dim=-1), rtol=1e-5)
    assert torch.equal(qo[:, :, 0], q[:, :, 0]) and torch.equal(ko[:, :, 0], k[:, :, 0])
    # same rotation for every head: rope(x) with x broadcast over heads is broadcast too
    x = q[:, :1].expand(B, QH, L, D).contiguous()
    xo, _ = pkg.llama_rope(x, k, cos=cos, sin=sin)
    assert torch.equal(xo[:, 0], xo[:, QH - 1])
    # bitwise reproducible
    qo2, _ = pkg.llama_rope(q, k, cos=cos, sin=sin)
    assert torch.equal(qo, qo2)


def test_feeds_flash_attention(pkg):
    """The caller chain of SURVEY.md 8(f): rope(q, k) -> flash_attention, gradients flow through both."""
    from oracle.naive_attention import naive_attention_grads
    B, QH, KH, L, D = 1, 4, 2, 256, 64
    q, k, cos, sin = _mk(11, B, QH, KH, L, D, "f32")
    v = torch.randn(B, KH, L, D, device=DEV)
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    qo, ko = pkg.llama_rope(q, k, cos=cos, sin=sin)
    o = pkg.flash_attention(qo, ko, v, causal=True)
    do = torch.randn_like(o)
    o.backward(do)
    rq, rk = naive_llama_rope(_np(q), _np(k), _np(cos), _np(sin))
    dq_r, dk_r, dv_r, _ = naive_attention_grads(rq, rk, _np(v), _np(do), None, causal=True, kpad_mask=None)
    gq, gk = pairwise_llama_rope(dq_r, dk_r, _np(cos), _np(sin), sin_sign=-1.0)
    for got, ref in ((q.grad, gq), (k.grad, gk), (v.grad, dv_r)):
        np.testing.assert_allclose(_np(got), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_host_checks(pkg):
    q, k, cos, sin = _mk(1, 2, 3, 1, 13, 16, "f32")
    with pytest.raises(pkg.NNopError, match="AssertionError"):
        pkg.llama_rope(q, k[:, :, :12].contiguous(), cos=cos, sin=sin)
    with pytest.raises(pkg.NNopError, match="shape"):
        pkg.llama_rope(q, k, cos=cos[:, :12].contiguous(), sin=sin)
    with pytest.raises(TypeError):
        pkg.llama_rope(q, k.half(), cos=cos, sin=sin)
    with pytest.raises(TypeError):
        pkg.llama_rope(q, k, cos=cos.half(), sin=sin.half())         # fp32 q with fp16 tables


import glob as _glob
import os as _os

_ROPE_GOLDEN = sorted(_glob.glob(_os.path.join(_os.path.dirname(__file__), "golden", "rope_*.npz")))


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("path", _ROPE_GOLDEN, ids=[_os.path.basename(p)[5:-4] for p in _ROPE_GOLDEN])
def test_rope_golden(pkg, path, dt):
    """Committed fixtures (bf16-exact inputs, fp64-oracle outputs): forward and pullback in all three dtypes."""
    g = np.load(path)
    t = lambda n: torch.tensor(g[n]).to(TORCH_DT[dt]).to(DEV)
    cos, sin = torch.tensor(g["cos"]).to(DEV), torch.tensor(g["sin"]).to(DEV)
    qo, ko = pkg._llama_rope(t("q"), t("k"), cos, sin, bwd=False)
    dq, dk = pkg.grad_llama_rope((t("dq_out"), t("dk_out")), cos, sin)
    for got, name in ((qo, "q_out"), (ko, "k_out"), (dq, "dq"), (dk, "dk")):
        _check(got, g[name].astype(np.float64), dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("which", ["q", "cos", "out"])
def test_offset_views_take_the_elementwise_kernel(pkg, dt, which):
    """A C / Julia caller may hand over a view that starts 4 or 8 bytes into an allocation: dense, but not 16-byte aligned.
    The launcher must not issue its 16-byte vector accesses there (csrc/rope.hip picks the element-wise kernel); results
    are the same as from aligned storage."""
    B, QH, KH, L, D = 2, 3, 2, 130, 64
    q, k, cos, sin = _mk(77, B, QH, KH, L, D, dt)
    ref_q, ref_k = pkg.llama_rope_into(torch.empty_like(q), torch.empty_like(k), q, k, cos, sin)

    def misaligned_like(t):
        flat = torch.empty(t.numel() + 2, dtype=t.dtype, device=t.device)
        view = flat[1:1 + t.numel()].view(t.shape)                  # one element (2 or 4 bytes) past a 16-byte boundary
        assert view.data_ptr() % 16 != 0 and view.is_contiguous()
        view.copy_(t)
        return view

    q_in, cos_in, sin_in = q, cos, sin
    q_out = torch.empty_like(q)
    if which == "q":
        q_in = misaligned_like(q)
    elif which == "cos":
        cos_in, sin_in = misaligned_like(cos), misaligned_like(sin)
    else:
        q_out = misaligned_like(q)
    got_q, got_k = pkg.llama_rope_into(q_out, torch.empty_like(k), q_in, k, cos_in, sin_in)
    torch.cuda.synchronize()
    assert torch.equal(got_q, ref_q) and torch.equal(got_k, ref_k)
