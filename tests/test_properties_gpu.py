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


def _fill_normal(t, seed):
    """N(0,1) into a tensor of more than 2^31 elements, one leading-dim slice at a time (bounded scratch)."""
    g = torch.Generator(device=t.device).manual_seed(seed)
    for i in range(t.shape[0]):
        t[i].copy_(torch.randn(t[i].shape, generator=g, device=t.device, dtype=torch.float32))


def test_attention_tensors_beyond_2g_elements(pkg, dev):
    """Sized for 288 GB: q, o, dO, dq hold 2^31 + 2^26 ELEMENTS each (4.1 GiB in bf16), so every (batch, head) offset past
    the first 2^31 elements needs 64-bit addressing.  The last (batch, kv-head) slice of the big launch -- forward and
    backward, GQA 2:1 -- must be bitwise what a launch on that slice alone produces; a middle slice too."""
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~36 GiB of device memory")
    dt = torch.bfloat16
    B, QH, KH, L, E = 33, 32, 16, 16384, 128              # 33 * 32 * 16384 * 128 = 2^31 + 2^26 elements
    assert B * QH * L * E > 2 ** 31
    q = torch.empty(B, QH, L, E, device=dev, dtype=dt); _fill_normal(q, 1)
    k = torch.empty(B, KH, L, E, device=dev, dtype=dt); _fill_normal(k, 2)
    v = torch.empty(B, KH, L, E, device=dev, dtype=dt); _fill_normal(v, 3)
    do = torch.empty(B, QH, L, E, device=dev, dtype=dt); _fill_normal(do, 4)
    o, ms, ls = pkg._flash_attention(q, k, v, causal=True)
    dq, dk, dv, _ = pkg.grad_flash_attention(do, o, ms, ls, q, k, v, causal=True)
    torch.cuda.synchronize()
    for b in (B - 1, B // 2):
        sl = slice(b, b + 1)
        o1, ms1, ls1 = pkg._flash_attention(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), causal=True)
        dq1, dk1, dv1, _ = pkg.grad_flash_attention(do[sl].contiguous(), o1, ms1, ls1, q[sl].contiguous(),
                                                    k[sl].contiguous(), v[sl].contiguous(), causal=True)
        torch.cuda.synchronize()
        for name, big, small in (("o", o, o1), ("ms", ms, ms1), ("ls", ls, ls1), ("dq", dq, dq1), ("dk", dk, dk1), ("dv", dv, dv1)):
            assert torch.equal(big[sl], small), f"{name} of batch {b}"
    assert torch.isfinite(o[-1].float()).all() and float(o[-1].float().abs().max()) > 0


def test_row_operators_beyond_2g_elements(pkg, dev):
    """The row-wise operators on a matrix of 2^31 + 2^20 elements: rows past element 2^31 equal a launch on those rows
    alone (bitwise), for softmax, both norms, and RoPE on a q tensor of that size."""
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~30 GiB of device memory")
    dt = torch.bfloat16
    n, emb = 2 ** 19 + 256, 4096
    assert n * emb > 2 ** 31
    x = torch.empty(n, emb, device=dev, dtype=dt)
    g = torch.Generator(device=dev).manual_seed(5)
    for i in range(0, n, 2 ** 16):
        x[i:i + 2 ** 16].copy_(torch.randn(x[i:i + 2 ** 16].shape, generator=g, device=dev))
    w = torch.randn(emb, device=dev, generator=g); b_ = torch.randn(emb, device=dev, generator=g)
    tail = slice(n - 300, n)
    xt = x[tail].contiguous()
    y = pkg.online_softmax(x)
    assert torch.equal(y[tail], pkg.online_softmax(xt))
    dx = pkg.grad_online_softmax(x, y)                          # any cotangent: reuse x
    assert torch.equal(dx[tail], pkg.grad_online_softmax(xt, y[tail].contiguous()))
    del dx
    yr, rms = pkg._rms_norm(x, w)
    yt, rmst = pkg._rms_norm(xt, w)
    assert torch.equal(yr[tail], yt) and torch.equal(rms[tail], rmst)
    yl, mu, sg = pkg._layer_norm(x, w, b_)
    ylt, mut, sgt = pkg._layer_norm(xt, w, b_)
    assert torch.equal(yl[tail], ylt) and torch.equal(mu[tail], mut) and torch.equal(sg[tail], sgt)
    # pullbacks: dx rows are independent of the other rows (dw / db are sums over ALL rows: compare to fp32 torch)
    dxr, dwr = pkg.grad_rms_norm(y, rms, x, w)
    dxt, _ = pkg.grad_rms_norm(y[tail].contiguous(), rmst, xt, w)
    assert torch.equal(dxr[tail], dxt)
    ref_dw = torch.zeros(emb, device=dev, dtype=torch.float64)
    for i in range(0, n, 2 ** 16):
        ref_dw += (y[i:i + 2 ** 16].double() * x[i:i + 2 ** 16].double() * rms[i:i + 2 ** 16, None].double()).sum(0)
    assert float((dwr.double() - ref_dw).abs().max() / ref_dw.abs().max()) < 1e-4
    del dxr, yr, yl, y
    # RoPE: q of > 2^31 elements, k small
    Bq, QHq, Lq, D = 1, 2 ** 11 + 1, 8192, 128
    q = x.view(-1)[: Bq * QHq * Lq * D].view(Bq, QHq, Lq, D)
    assert q.numel() > 2 ** 31
    kk = torch.randn(Bq, 1, Lq, D, device=dev, generator=g).to(dt)
    cos, sin = pkg.LlamaRotaryEmbedding(D)(torch.arange(Lq, device=dev, dtype=torch.float32)[None])
    qo, ko = pkg.llama_rope(q, kk, cos=cos, sin=sin)
    qo1, _ = pkg.llama_rope(q[:, -1:].contiguous(), kk, cos=cos, sin=sin)
    assert torch.equal(qo[:, -1:], qo1)
