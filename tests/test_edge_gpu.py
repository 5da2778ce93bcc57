"""GPU parity on edge shapes the reference's grids do not reach: single query / single key, QL != KL with
causal (top-left aligned), one KV head shared by all query heads, lengths around the tile sizes (32, 64, 128, 224,
256), batch/head = 1, E at both ends of the supported range."""
import pytest
import torch

from util import make_inputs, oracle_fwd, oracle_bwd, assert_close

pytestmark = pytest.mark.gpu

SHAPES = [
    # B, QH, KH, QL, KL
    (1, 1, 1, 1, 1), (1, 1, 1, 1, 77), (1, 1, 1, 77, 1), (2, 3, 1, 5, 3), (1, 2, 2, 33, 31),
    (1, 1, 1, 64, 64), (1, 1, 1, 65, 63), (2, 2, 1, 129, 127), (1, 4, 2, 225, 223), (1, 1, 1, 257, 449),
    (1, 1, 1, 31, 300), (1, 8, 1, 96, 160),
]


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("E", [16, 128])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("B,QH,KH,QL,KL", SHAPES)
def test_edge_shapes_fwd_bwd(pkg, dev, dt, E, causal, B, QH, KH, QL, KL):
    d = make_inputs(41, B, QH, KH, QL, KL, E, dt, dev)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal)
    dq, dk, dv, _ = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal)
    torch.cuda.synchronize()
    o_ref, ms_ref, _ = oracle_fwd(d, causal)
    rq, rk, rv, _ = oracle_bwd(d, causal)
    assert_close("o", o, o_ref, dt)
    assert_close("ms", ms, ms_ref, dt)
    sc = 1.0 if dt == "f32" else 2.0
    assert_close("dq", dq, rq, dt, sc, floor=True)
    assert_close("dk", dk, rk, dt, sc, floor=True)
    assert_close("dv", dv, rv, dt, sc, floor=True)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
def test_adversarial_logits_force_the_rescale_path(pkg, dev, dt):
    """Deferred-max softmax: spike single (query, key) pairs so that a row's max jumps far past the 2^8 threshold in
    the middle of the key sweep, in different tiles for different rows, and make some rows extremely negative
    (reference max adopted at the first visible key).  A rescale bug shows up as O(0.1) errors (no NaN)."""
    B, H, L, E = 1, 2, 640, 64
    d = make_inputs(43, B, H, H, L, L, E, dt, dev)
    q, k = d["q"].float(), d["k"].float()
    for row, key, gain in [(5, 70, 9.0), (5, 400, 20.0), (100, 639, 30.0), (333, 200, 14.0), (600, 3, 25.0)]:
        k[0, :, key] = q[0, :, row] * gain / q[0, :, row].norm(dim=-1, keepdim=True) * (E ** 0.5) / 3
    q[0, 1, 50:60] *= 40.0                      # huge logits, both signs
    k[0, 1, :5] = -k[0, 1, :5].abs() * 3
    d["q"], d["k"] = q.to(d["v"].dtype), k.to(d["v"].dtype)
    for causal in (False, True):
        o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal)
        g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal)
        torch.cuda.synchronize()
        o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
        assert torch.isfinite(o.float()).all()
        assert_close("o", o, o_ref, dt, 2.0)
        assert_close("ms", ms, ms_ref, dt, 2.0)
        rq, rk, rv, _ = oracle_bwd(d, causal)
        if dt == "f32":
            assert_close("dq", g[0], rq, dt, 2.0)
            assert_close("dk", g[1], rk, dt, 2.0)
            assert_close("dv", g[2], rv, dt, 2.0)
        else:
            # 16-bit: dS is rounded to T before the dK / dQ MFMAs, and the rows scaled by 40 amplify that rounding
            # unit element-wise; the gradients are checked norm-wise (measured: bf16 <= 2.1e-2, f16 <= 3.4e-3)
            import numpy as np
            for name, got, ref in (("dq", g[0], rq), ("dk", g[1], rk), ("dv", g[2], rv)):
                x = got.double().cpu().numpy()
                assert np.isfinite(x).all()
                assert np.linalg.norm(x - ref) <= (5e-2 if dt == "bf16" else 1e-2) * np.linalg.norm(ref), name


@pytest.mark.parametrize("dt,causal", [("bf16", False), ("f32", True)])       # (the two other combinations ran green through round 2; suite time)
def test_key_padding_beyond_the_lds_validity_words(pkg, dev, dt, causal):
    """Key-padding masks are turned into one 64-bit validity word per 64-key tile in LDS for up to 1024 tiles
    (kMaxMaskTiles); longer key sequences read the mask per tile instead.  KL = 65536 + 200 crosses that limit: one batch
    is valid almost to the end (the tiles past the limit are live), one stops early, one has holes on both sides of it."""
    import numpy as np
    B, QH, KH, QL, KL, E = 3, 2, 1, 96, 65536 + 200, 64
    d = make_inputs(77, B, QH, KH, QL, KL, E, dt, dev)
    m = np.zeros((B, KL), dtype=bool)
    m[0, : KL - 7] = True
    m[1, :40000] = True
    m[2] = np.random.default_rng(3).random(KL) < 0.5
    m[2, 65500:65600] = False
    m[2, 65700] = True
    d["mask"] = torch.tensor(m).to(dev)
    if causal:
        # causal needs queries at the far end to see the long tail: QL == KL is too big for the oracle, so shift the
        # valid keys instead -- with top-left alignment query i sees keys <= i, i.e. only the first 96 keys here
        d["mask"][:, :4] = True
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
    dq, dk, dv, _ = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
    torch.cuda.synchronize()
    o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
    assert_close("o", o, o_ref, dt, floor=True)
    lse = ms.double().cpu().numpy() + np.log(ls.double().cpu().numpy())
    assert_close("lse", lse, ms_ref + np.log(ls_ref), dt, floor=True)
    rq, rk, rv, _ = oracle_bwd(d, causal)
    sc = 1.0 if dt == "f32" else 2.0
    assert_close("dq", dq, rq, dt, sc, floor=True)
    assert_close("dk", dk, rk, dt, sc, floor=True)
    assert_close("dv", dv, rv, dt, sc, floor=True)
    # masked keys get exactly zero gradient
    dead = ~d["mask"]
    assert (dk[dead[:, None, :].expand(B, KH, KL)] == 0).all() and (dv[dead[:, None, :].expand(B, KH, KL)] == 0).all()


@pytest.mark.parametrize("dt", ["bf16", "f32"])
@pytest.mark.parametrize("causal", [False, True])
def test_masked_keys_with_overflowing_values_do_not_leak(pkg, dev, dt, causal):
    """Padded-out keys carry huge K / V values (their unmasked logits overflow exp2 to +inf).  The dK/dV kernel lets a
    masked key accumulate whatever it likes in its OWN lane and zeroes that lane before the store, so nothing may leak:
    every output and gradient is finite, equals the oracle, and the masked keys' dK / dV are exactly zero."""
    import numpy as np
    B, QH, KH, QL, KL, E = 2, 4, 2, 200, 333, 64
    d = make_inputs(91, B, QH, KH, QL, KL, E, dt, dev)
    m = np.random.default_rng(7).random((B, KL)) < 0.6
    m[:, 0] = True
    m[1, 100:180] = False                                    # a whole 64-key tile of dead keys inside
    d["mask"] = torch.tensor(m).to(dev)
    dead = ~d["mask"]
    big = 3.0e4 if dt == "bf16" else 1.0e6
    with torch.no_grad():
        d["k"][dead[:, None, :].expand(B, KH, KL)] = big     # q.k * scale ~ +-1e4 * sqrt(E): exp2 overflows
        d["v"][dead[:, None, :].expand(B, KH, KL)] = -big
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
    dq, dk, dv, _ = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
    torch.cuda.synchronize()
    for t in (o, dq, dk, dv):
        assert torch.isfinite(t.float()).all()
    o_ref, _, _ = oracle_fwd(d, causal)
    rq, rk, rv, _ = oracle_bwd(d, causal)
    sc = 1.0 if dt == "f32" else 2.0
    assert_close("o", o, o_ref, dt, floor=True)
    assert_close("dq", dq, rq, dt, sc, floor=True)
    live = d["mask"][:, None, :, None].expand(B, KH, KL, E).cpu().numpy()
    assert_close("dk", np.where(live, dk.double().cpu().numpy(), 0.0), np.where(live, rk, 0.0), dt, sc, floor=True)
    assert_close("dv", np.where(live, dv.double().cpu().numpy(), 0.0), np.where(live, rv, 0.0), dt, sc, floor=True)
    assert (dk[dead[:, None, :].expand(B, KH, KL)] == 0).all() and (dv[dead[:, None, :].expand(B, KH, KL)] == 0).all()


@pytest.mark.parametrize("dt,E", [("f32", 64), ("f32", 32), ("bf16", 16)])
@pytest.mark.parametrize("QH,KH", [(8, 8), (8, 4)])
def test_causal_block_order_of_the_32_row_forward_does_not_change_results(pkg, dev, tune, dt, E, QH, KH):
    """csrc/fa_fwd_inst.hpp launch_fwd_cfg: under a causal mask every second column of an XCD's dispatch order may run its q-blocks
    ascending (heavy blocks beside light ones on a CU; knob fwd_causal_alt) -- a permutation of which workgroup does which block, so the
    outputs are bitwise those of the plain heaviest-first order, and right"""
    d = make_inputs(131, 2, QH, KH, 1100, 1100, E, dt, dev, need_do=False)      # 16 columns: 2 per XCD (the chunked remap applies)
    outs = []
    for alt in (0, 1):
        tune(fwd_causal_alt=alt, fwd_duo=0, fwd_w64=0)
        o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=True)
        torch.cuda.synchronize()
        outs.append((o, ms, ls))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    o_ref, ms_ref, _ = oracle_fwd(d, True)
    assert_close("o", outs[1][0], o_ref, dt)
    assert_close("ms", outs[1][1], ms_ref, dt)
    # the 32-row backward kernels take the same switch (fp32, and 16-bit E <= 32 -- no one-wave-per-SIMD form there)
    do = torch.randn_like(d["q"])
    grads = []
    for alt in (0, 1):
        tune(fwd_causal_alt=alt, fwd_duo=0, fwd_w64=0, bwd_w64=0)
        o, ms, ls = outs[0]
        grads.append(pkg.grad_flash_attention(do, o, ms, ls, d["q"], d["k"], d["v"], None, causal=True)[:3])
        torch.cuda.synchronize()
    for a, b in zip(*grads):
        assert torch.equal(a, b)
