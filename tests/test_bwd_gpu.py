"""GPU parity: HIP backward (through the C ABI) vs the fp64 oracle's analytic gradients, with a
random cotangent dO (stronger than the reference's sum(o), test/attention_tests.jl:36-41).
Grids mirror the reference's at reduced H,B plus fp16/bf16 and E=128."""
import pytest
import torch

from util import make_inputs, oracle_bwd, assert_close

pytestmark = pytest.mark.gpu


def check_bwd(pkg, d, causal, dt):
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    dq, dk, dv, dp = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], d["pair"],
                                              causal=causal, kpad_mask=d["mask"])
    torch.cuda.synchronize()
    rq, rk, rv, rp = oracle_bwd(d, causal)
    # 16-bit residuals (ms, ls in T, src/attention.jl:167-168) add a row-uniform relative error of one
    # T-ulp to P on top of the operand rounding: allow 2x the forward tolerance for the gradients.
    sc = 1.0 if dt == "f32" else 2.0
    assert_close("dq", dq, rq, dt, sc)
    assert_close("dk", dk, rk, dt, sc)
    assert_close("dv", dv, rv, dt, sc)
    if d["pair"] is not None:
        assert_close("dpair", dp, rp, dt, sc)
    else:
        assert dp is None


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("E", [16, 32, 64, 128])
@pytest.mark.parametrize("QL,KL", [(255, 255), (256, 512), (511, 256), (512, 1024)])
def test_noncausal(pkg, dev, dt, E, QL, KL):
    d = make_inputs(11, 2, 2, 2, QL, KL, E, dt, dev)
    check_bwd(pkg, d, False, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("E", [16, 64, 128])
@pytest.mark.parametrize("L", [255, 256, 511, 1024])
@pytest.mark.parametrize("pad", [None, "ref"])
def test_causal(pkg, dev, dt, E, L, pad):
    d = make_inputs(12, 2, 2, 2, L, L, E, dt, dev, pad=pad)
    check_bwd(pkg, d, True, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("QH,KH", [(4, 1), (6, 2), (8, 2)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("L", [255, 257, 512])
def test_gqa(pkg, dev, dt, QH, KH, causal, L):
    d = make_inputs(13, 2, QH, KH, L, L, 32, dt, dev)
    check_bwd(pkg, d, causal, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("pad", ["ref", "lens", "random"])
@pytest.mark.parametrize("causal", [False, True])
def test_padmask(pkg, dev, dt, pad, causal):
    d = make_inputs(14, 3, 2, 2, 700, 700, 64, dt, dev, pad=pad)
    check_bwd(pkg, d, causal, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("pad", [None, "ref"])
@pytest.mark.parametrize("E", [32, 64])
def test_pair_and_dpair(pkg, dev, dt, causal, pad, E):
    d = make_inputs(15, 2, 2, 2, 300, 300, E, dt, dev, pair=True, pad=pad)
    check_bwd(pkg, d, causal, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("causal,pad", [(False, None), (True, "lens"), (False, "random")])
@pytest.mark.parametrize("E,QH,KH,QL,KL", [(64, 4, 4, 300, 300), (32, 6, 2, 257, 190), (128, 2, 1, 100, 333), (16, 3, 3, 64, 64)])
def test_pair_backward_staged_and_direct_paths(pkg, dev, dt, causal, pad, E, QH, KH, QL, KL):
    """The backward of a call with a pair bias has two paths behind one entry point: with the scratch of
    nnop_fa_bwd_workspace_bytes_pair it re-packs the bias head-major and works on 16-byte accesses (csrc/pair_tile.hpp), with the
    small workspace it touches pair / dpair element-wise.  Both against the oracle, and dpair of the two bitwise equal (the same
    products in the same order; only the route of the bias and of dS differs)."""
    d = make_inputs(17, 2, QH, KH, QL, KL, E, dt, dev, pair=True, pad=pad)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    small = pkg.bwd_workspace_bytes(d["q"], d["k"], d["v"], causal=causal)
    big = pkg.bwd_workspace_bytes(d["q"], d["k"], d["v"], causal=causal, pair=True)
    assert big > small
    outs = []
    for nbytes in (small, big):
        dq, dk, dv, dp = (torch.full_like(d[n], float("nan")) for n in ("q", "k", "v", "pair"))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        pkg.fa_bwd_into(dq, dk, dv, dp, ws, d["do"], o, ms, ls, d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        outs.append((dq, dk, dv, dp))
    rq, rk, rv, rp = oracle_bwd(d, causal)
    for dq, dk, dv, dp in outs:
        assert_close("dq", dq, rq, dt, kind="grad")
        assert_close("dk", dk, rk, dt, kind="grad")
        assert_close("dv", dv, rv, dt, kind="grad")
        assert_close("dpair", dp, rp, dt, kind="grad")
    assert torch.equal(outs[0][3], outs[1][3]), "dpair differs between the direct and the staged path"


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [16, 32, 64])
@pytest.mark.parametrize("case", ["plain", "causal", "pad", "gqa", "pair"])
def test_eight_wave_backward_form(pkg, dev, tune, dt, E, case):
    """16-bit E <= 64: the launcher uses 8-wave workgroups (256 keys / queries sharing each staged tile) for non-causal problems
    that still fill the chip (csrc/fa_bwd_inst.hpp, kWide8).  Forced here on small shapes in every mode, ragged lengths included,
    against the oracle and bitwise against the 4-wave form (same products per (query, key) pair in the same order)."""
    QH, KH = (6, 2) if case == "gqa" else (2, 2)
    d = make_inputs(18, 2, QH, KH, 517, 390, E, dt, dev, pair=(case == "pair"), pad=("lens" if case == "pad" else None))
    causal = case == "causal"
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    outs = []
    for nw in (4, 8):
        tune(bwd_nw=nw)
        outs.append(pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"]))
    torch.cuda.synchronize()
    rq, rk, rv, rp = oracle_bwd(d, causal)
    for dq, dk, dv, dp in outs:
        assert_close("dq", dq, rq, dt, kind="grad")
        assert_close("dk", dk, rk, dt, kind="grad")
        assert_close("dv", dv, rv, dt, kind="grad")
        if dp is not None:
            assert_close("dpair", dp, rp, dt, kind="grad")
    assert torch.equal(outs[0][0], outs[1][0]), "dq differs between the 4-wave and the 8-wave form"


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_rrule_through_autograd(pkg, dev, dt):
    """src/attention_crc.jl:16-31: the pullback of flash_attention returns (dq, dk, dv, dpair)."""
    d = make_inputs(16, 2, 4, 2, 200, 200, 64, dt, dev, pair=True, pad="ref")
    q, k, v, pair = (d[n].clone().requires_grad_(True) for n in ("q", "k", "v", "pair"))
    o = pkg.flash_attention(q, k, v, pair, causal=True, kpad_mask=d["mask"])
    o.backward(d["do"])
    torch.cuda.synchronize()
    rq, rk, rv, rp = oracle_bwd(d, True)
    sc = 1.0 if dt == "f32" else 2.0
    assert_close("dq", q.grad, rq, dt, sc)
    assert_close("dk", k.grad, rk, dt, sc)
    assert_close("dv", v.grad, rv, dt, sc)
    assert_close("dpair", pair.grad, rp, dt, sc)
    with torch.no_grad():      # not under AD: plain forward, returns o only (attention_crc.jl:11-13)
        o2 = pkg.flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=True, kpad_mask=d["mask"])
    assert torch.equal(o2, o.detach())


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("causal", [False, True])
def test_e128_seven_wave_backward_form(pkg, dev, dt, causal, tune):
    """E = 128 backward has two workgroup shapes (4 waves / 7 waves with single-buffered tiles, chosen by grid
    size).  Force each and check both against the oracle; 224-row blocks exercise ragged block tails."""
    d = make_inputs(17, 2, 4, 2, 700, 700, 128, dt, dev, pad="ref" if causal else None)
    for thr in (1, 100000000):
        tune(bwd_big7=thr)
        check_bwd(pkg, d, causal, dt)


@pytest.mark.parametrize("causal,pad,pair", [(False, None, False), (True, None, False), (False, "lens", False), (True, "random", False),
                                             (False, "ref", True), (True, None, True)])
def test_f32_e128_four_wave_register_form(pkg, dev, causal, pad, pair):
    """fp32 E = 128 backward (csrc/fa_bwd.hpp, fa_bwd_f32_wide): 4 waves per workgroup, one per SIMD, K fragments (dK/dV) / Q and dO
    fragments (dQ) in registers beside the accumulators -- the > 256-register mode.  Several 128-key / 128-query blocks, GQA 4/2,
    ragged lengths; against the oracle, and bitwise equal over repeated launches with the caches flushed in between (no atomics:
    a hazard or race in the wide register allocation would show here, a tolerance check can miss it)."""
    d = make_inputs(61, 2, 4, 2, 709, 645, 128, "f32", dev, pad=pad, pair=pair)
    check_bwd(pkg, d, causal, "f32")
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    runs = []
    for _ in range(3):
        flush.fill_(1)
        g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        runs.append(g)
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            if a is not None:
                assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
