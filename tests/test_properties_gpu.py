"""Size-independent properties at BASELINE.json's FULL sizes (where the fp64 oracle is too slow):
linearity in V, softmax normalisation, causality, GQA == repeated KV heads, batch/head independence
(the sharding argument of SURVEY.md section 8(e)), key-permutation invariance, and agreement of the
bf16 path with the fp32 path of the same library."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def randn(dev, dt, *s, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.randn(*s, generator=g, device=dev, dtype=torch.float32).to(dt)


def relmax(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


# C2: bf16 non-causal E=64 L=4096 H=4 B=4 (headline); C3-shaped but smaller batch: causal E=128 L=8192
FULL = [("c2", torch.bfloat16, 4, 4, 4, 4096, 64, False), ("c3", torch.bfloat16, 1, 8, 8, 8192, 128, True),
        ("c4", torch.float16, 2, 32, 8, 4096, 128, False)]


@pytest.mark.parametrize("name,dt,B,QH,KH,L,E,causal", FULL, ids=[f[0] for f in FULL])
def test_full_size_properties(pkg, dev, name, dt, B, QH, KH, L, E, causal):
    q, k = randn(dev, dt, B, QH, L, E, seed=1), randn(dev, dt, B, KH, L, E, seed=2)
    v1, v2 = randn(dev, dt, B, KH, L, E, seed=3), randn(dev, dt, B, KH, L, E, seed=4)
    fa = lambda vv, **kw: pkg.flash_attention(q, k, vv, causal=causal, **kw)
    o1, o2 = fa(v1), fa(v2)
    # (1) softmax rows sum to one: V = ones -> O = ones (bf16: P rounded to T before the PV product)
    ones = torch.ones_like(v1)
    assert relmax(fa(ones), torch.ones_like(o1)) < 1e-2
    # (2) linearity in V (exact arithmetic identity; rounding of v1+v2 and of outputs only)
    vs = (v1.float() * 0.5 + v2.float() * 0.25).to(dt)
    lin = o1.float() * 0.5 + o2.float() * 0.25
    assert relmax(fa(vs), lin) < 2e-2
    # (3) causality: changing keys/values at positions > t does not change rows <= t
    if causal:
        t = L // 2 + 17
        k2, v3 = k.clone(), v1.clone()
        k2[:, :, t + 1:] = randn(dev, dt, B, KH, L - t - 1, E, seed=9)
        v3[:, :, t + 1:] = 0
        o3 = pkg.flash_attention(q, k2, v3, causal=True)
        assert torch.equal(o3[:, :, :t + 1], o1[:, :, :t + 1])
        # first row attends only key 0
        rep = QH // KH
        assert relmax(o1[:, :, 0], v1.repeat_interleave(rep, dim=1)[:, :, 0]) < 1e-2
    # (4) GQA == attention with the KV heads repeated (test/attention_testsetup.jl:23-30)
    if QH != KH:
        rep = QH // KH
        o_rep = pkg.flash_attention(q, k.repeat_interleave(rep, 1), v1.repeat_interleave(rep, 1), causal=causal)
        assert torch.equal(o_rep, o1)
    # (5) (batch, head) slices are independent -> sharding them needs no collective
    b, h = B - 1, QH - 1
    kh = h // (QH // KH)
    o_s = pkg.flash_attention(q[b:b + 1, h:h + 1], k[b:b + 1, kh:kh + 1], v1[b:b + 1, kh:kh + 1], causal=causal)
    assert relmax(o_s, o1[b:b + 1, h:h + 1]) < (1e-6 if dt == torch.float32 else 1e-2)
    # (6) outputs are convex combinations of V rows: bounded by max |v|
    assert float(o1.float().abs().max()) <= float(v1.float().abs().max()) * (1 + 1e-2)
    assert torch.isfinite(o1.float()).all()


def test_key_permutation_invariance_and_padding_equivalence(pkg, dev):
    """Non-causal attention is invariant under a permutation of the keys; masking the last n keys
    equals dropping them (variable sequence length, README.md:48-50)."""
    dt = torch.bfloat16
    B, H, L, E = 2, 4, 4096, 64
    q, k, v = randn(dev, dt, B, H, L, E, seed=1), randn(dev, dt, B, H, L, E, seed=2), randn(dev, dt, B, H, L, E, seed=3)
    o = pkg.flash_attention(q, k, v, causal=False)
    perm = torch.randperm(L, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    o_p = pkg.flash_attention(q, k[:, :, perm], v[:, :, perm], causal=False)
    assert relmax(o_p, o) < 1e-2
    n = 1000
    mask = torch.ones(B, L, dtype=torch.bool, device=dev)
    mask[:, L - n:] = False
    o_m = pkg.flash_attention(q, k, v, causal=False, kpad_mask=mask)
    o_d = pkg.flash_attention(q, k[:, :, :L - n].contiguous(), v[:, :, :L - n].contiguous(), causal=False)
    assert relmax(o_m, o_d) < 1e-2


def test_bf16_path_agrees_with_fp32_path_at_headline_size(pkg, dev):
    """Forward and backward at C2: the bf16 kernels against the exact-fp32-MFMA kernels of the same
    library on the same (bf16-rounded) inputs."""
    B, H, L, E = 4, 4, 4096, 64
    mk = lambda s: randn(dev, torch.bfloat16, B, H, L, E, seed=s)
    q, k, v, do = mk(1), mk(2), mk(3), mk(4)
    outs = {}
    for dt in (torch.bfloat16, torch.float32):
        a, b, c, d = (t.to(dt) for t in (q, k, v, do))
        o, ms, ls = pkg._flash_attention(a, b, c, causal=False)
        g = pkg.grad_flash_attention(d, o, ms, ls, a, b, c, causal=False)
        outs[dt] = (o, *g[:3])
    for x, y, name in zip(outs[torch.bfloat16], outs[torch.float32], ("o", "dq", "dk", "dv")):
        assert relmax(x, y) < 2e-2, name


def test_gradient_linearity_in_cotangent_full_size(pkg, dev):
    """The pullback is linear in dO (src/attention_crc.jl:24-29)."""
    dt = torch.bfloat16
    B, QH, KH, L, E = 1, 8, 2, 4096, 128
    q, k, v = randn(dev, dt, B, QH, L, E, seed=1), randn(dev, dt, B, KH, L, E, seed=2), randn(dev, dt, B, KH, L, E, seed=3)
    d1, d2 = randn(dev, dt, B, QH, L, E, seed=4), randn(dev, dt, B, QH, L, E, seed=5)
    o, ms, ls = pkg._flash_attention(q, k, v, causal=True)
    g = lambda d: pkg.grad_flash_attention(d, o, ms, ls, q, k, v, causal=True)[:3]
    g1, g2, g3 = g(d1), g(d2), g((d1.float() - d2.float() * 0.5).to(dt))
    for a, b, c, name in zip(g1, g2, g3, ("dq", "dk", "dv")):
        assert relmax(c, a.float() - 0.5 * b.float()) < 3e-2, name
