"""GPU parity: HIP forward (through the C ABI) vs the fp64 oracle on the same seeded inputs.
Mirrors the reference's grids (test/attention_tests.jl:6-20, causal_attention_tests.jl:6-18,
gqa_attention_tests.jl:6-19) at reduced H,B plus fp16/bf16 and E=128."""
import numpy as np
import pytest
import torch

from util import make_inputs, oracle_fwd, assert_close

pytestmark = pytest.mark.gpu


def run_fwd(pkg, d, causal):
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    torch.cuda.synchronize()
    return o, ms, ls


def check_fwd(pkg, d, causal, dt):
    o, ms, ls = run_fwd(pkg, d, causal)
    o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
    assert_close("o", o, o_ref, dt)
    # residual contract: ms = row max, ls = sum exp(s - ms); compare the invariant ms + log(ls)
    lse = ms.double().cpu().numpy() + np.log(ls.double().cpu().numpy())
    with np.errstate(divide="ignore", invalid="ignore"):
        lse_ref = ms_ref + np.log(ls_ref)
    assert_close("lse", lse, lse_ref, dt)
    assert_close("ms", ms, ms_ref, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("E", [16, 32, 64, 128])
@pytest.mark.parametrize("QL,KL", [(255, 255), (256, 512), (511, 256), (512, 1024), (1024, 255)])
def test_noncausal(pkg, dev, dt, E, QL, KL):
    d = make_inputs(1, 2, 2, 2, QL, KL, E, dt, dev, need_do=False)
    check_fwd(pkg, d, False, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("E", [16, 64, 128])
@pytest.mark.parametrize("L", [255, 256, 511, 1024])
@pytest.mark.parametrize("pad", [None, "ref"])
def test_causal(pkg, dev, dt, E, L, pad):
    d = make_inputs(2, 2, 2, 2, L, L, E, dt, dev, pad=pad, need_do=False)
    check_fwd(pkg, d, True, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("QH,KH", [(4, 1), (6, 2), (8, 2)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("L", [255, 257, 512])
def test_gqa(pkg, dev, dt, QH, KH, causal, L):
    d = make_inputs(3, 2, QH, KH, L, L, 64, dt, dev, need_do=False)
    check_fwd(pkg, d, causal, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("pad", ["ref", "lens", "random"])
@pytest.mark.parametrize("causal", [False, True])
def test_padmask(pkg, dev, dt, pad, causal):
    d = make_inputs(4, 3, 2, 2, 700, 700, 64, dt, dev, pad=pad, need_do=False)
    check_fwd(pkg, d, causal, dt)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("pad", [None, "ref"])
def test_pair(pkg, dev, dt, causal, pad):
    d = make_inputs(5, 2, 2, 2, 300, 300, 32, dt, dev, pair=True, pad=pad, need_do=False)
    check_fwd(pkg, d, causal, dt)
