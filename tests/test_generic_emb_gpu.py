"""GPU parity for the embedding dims outside the MFMA kernels' set {16, 32, 64, 128}: the reference accepts any power of two
that fits shared memory (src/attention.jl:143,193-205); here 1 .. 512 run through csrc/fa_generic.hpp (plain HIP, one wave per
row).  Forward (o, ms, ls) and backward (dq, dk, dv, dpair) against the fp64 oracle in every mode."""
import pytest
import torch

from util import make_inputs, oracle_fwd, oracle_bwd, assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("E", [1, 2, 8, 256, 512])
@pytest.mark.parametrize("case", ["plain", "causal", "pad", "gqa", "pair", "causal_pad_pair"])
def test_forward_and_backward(pkg, dev, dt, E, case):
    QH, KH = (6, 2) if case == "gqa" else (2, 2)
    pad = "lens" if "pad" in case else None
    d = make_inputs(19, 2, QH, KH, 77, 130, E, dt, dev, pair=("pair" in case), pad=pad)
    causal = "causal" in case
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    dq, dk, dv, dp = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    torch.cuda.synchronize()
    o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
    assert_close("o", o, o_ref, dt)
    assert_close("ms", ms, ms_ref, dt)
    rq, rk, rv, rp = oracle_bwd(d, causal)
    assert_close("dq", dq, rq, dt, kind="grad")
    assert_close("dk", dk, rk, dt, kind="grad")
    assert_close("dv", dv, rv, dt, kind="grad")
    if dp is not None:
        assert_close("dpair", dp, rp, dt, kind="grad")


def test_through_the_public_entry_point(pkg, dev):
    """flash_attention + autograd at E = 256 (the head dim the MFMA kernels do not cover)."""
    d = make_inputs(20, 1, 4, 4, 200, 200, 256, "bf16", dev)
    q, k, v = (d[n].clone().requires_grad_(True) for n in ("q", "k", "v"))
    o = pkg.flash_attention(q, k, v, causal=True)
    o.backward(d["do"])
    torch.cuda.synchronize()
    rq, rk, rv, _ = oracle_bwd(d, True)
    assert_close("dq", q.grad, rq, "bf16", kind="grad")
    assert_close("dk", k.grad, rk, "bf16", kind="grad")
    assert_close("dv", v.grad, rv, "bf16", kind="grad")


@pytest.mark.parametrize("dt", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("causal", [False, True])
def test_e256_tiled_kernels_larger_shape_and_reproducibility(pkg, dev, dt, causal):
    """E = 256 on the MFMA kernels -- 16-bit: the one-wave-per-SIMD forms (csrc/fa_fwd_w64.hpp, fa_bwd_w64.hpp); fp32: the 32-row tiled
    FORWARD with the whole register file per wave (its backward is the plain-HIP path, fed with the tiled forward's residuals): a
    multi-tile, multi-workgroup shape with GQA and a ragged key length, and bitwise-equal repeat launches."""
    d = make_inputs(21, 2, 4, 2, 389, 517, 256, dt, dev, pad="lens")
    outs = []
    for _ in range(2):
        o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
        g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        outs.append((o, ms, ls, *g[:3]))
    for a, b in zip(*outs):
        assert torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
    o_ref, ms_ref, _ = oracle_fwd(d, causal)
    rq, rk, rv, _ = oracle_bwd(d, causal)
    o, ms, ls, dq, dk, dv = outs[0]
    assert_close("o", o, o_ref, dt)
    assert_close("ms", ms, ms_ref, dt)
    assert_close("dq", dq, rq, dt, kind="grad")
    assert_close("dk", dk, rk, dt, kind="grad")
    assert_close("dv", dv, rv, dt, kind="grad")
