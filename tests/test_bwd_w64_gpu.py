"""GPU parity of the one-wave-per-SIMD backward (csrc/fa_bwd_w64.hpp: dK/dV and dQ as one kernel template with inline-asm MFMAs,
LDS-DMA into dual-use images, a hand-placed three-stage pipeline) against the fp64 oracle's analytic gradients
(src/attention_bwd.jl:86-156) -- forced through the debug hook so that small shapes reach it -- plus bitwise reproducibility
across launches (the screen for races of the DMA ring and for reads of MFMA results that have not landed) and agreement with
the 32-row kernels of csrc/fa_bwd.hpp on the same residuals."""
import numpy as np
import pytest
import torch

from util import make_inputs, oracle_bwd, assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1], ids=["rows64", "rows32"])
def shape(request, tune):
    """both wave shapes of the kernel template on every case of this file: 64 stationary rows per wave (256-row workgroups) and the
    narrow one (BwdW64Shape NARROW: 32 rows per wave, 64 streamed rows per step, 128-row workgroups) that the launcher picks where
    256-row blocks would leave CUs idle (knob bwd_narrow; tests below for the rule itself)"""
    # suite time: fp16 and bf16 run the same kernel templates (they differ in the MFMA opcode); the narrow shape runs its fp16 repeats
    # only where a test has no bf16 twin
    dt = request.node.callspec.params.get("dt") if hasattr(request.node, "callspec") else None
    if request.param == 1 and dt == "f16" and "test_noncausal" not in request.node.name:
        pytest.skip("narrow shape: fp16 repeat of a bf16 case")
    tune(bwd_narrow=request.param)
    return request.param


def run_bwd(pkg, d, causal):
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])
    dq, dk, dv, dp = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])
    torch.cuda.synchronize()
    assert dp is None
    return dq, dk, dv


def check(pkg, d, causal, dt, floor=False):
    dq, dk, dv = run_bwd(pkg, d, causal)
    rq, rk, rv, _ = oracle_bwd(d, causal)
    assert_close("dv", dv, rv, dt, kind="grad", floor=floor)
    assert_close("dk", dk, rk, dt, kind="grad", floor=floor)
    assert_close("dq", dq, rq, dt, kind="grad", floor=floor)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("which", [1, 2, 3, 4])
@pytest.mark.parametrize("QL,KL", [(64, 64), (255, 257), (512, 1024), (1100, 300), (33, 1000), (1, 1)])
def test_noncausal(pkg, dev, tune, dt, E, which, QL, KL):
    """which: 1 both passes on the new form (the dQ kernel computes the row constants: no preprocess launch), 2 dK/dV only, 3 dQ only
    (the other pass on csrc/fa_bwd.hpp), 4 both with the separate preprocess launch"""
    # (tests/test_bwd_gpu.py's structured grid -- E in {16 .. 128} x 4 shapes x 3 dtypes, causal x key padding -- runs through these
    # kernels by default as well; this file adds what that grid lacks)
    if which != 1 and (QL, KL) not in ((255, 257), (512, 1024)):
        pytest.skip("the mixed combinations on two shapes only")
    if which == 1 and (QL, KL) in ((255, 257), (512, 1024)) and dt == "f16":
        pytest.skip("covered in bf16 and by test_bwd_gpu.py")
    tune(bwd_w64=which)
    # a single key: dS = P (dP - delta) cancels to exactly zero in the oracle (tests/util.py, ABS_FLOOR)
    check(pkg, make_inputs(91, 2, 2, 2, QL, KL, E, dt, dev), False, dt, floor=(KL == 1))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("L", [64, 257, 700])
@pytest.mark.parametrize("pad", [None, "ref"])
def test_causal(pkg, dev, tune, dt, E, L, pad):
    if dt == "f16" and pad == "ref":
        pytest.skip("covered in bf16")
    tune(bwd_w64=1)
    check(pkg, make_inputs(92, 2, 2, 2, L, L, E, dt, dev, pad=pad), True, dt)


@pytest.mark.parametrize("dt", ["bf16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("QL,KL", [(300, 700), (700, 300)])
def test_causal_rectangular(pkg, dev, tune, dt, E, QL, KL):
    """top-left aligned causal mask for QL != KL (deviation 4)"""
    tune(bwd_w64=1)
    check(pkg, make_inputs(93, 2, 2, 2, QL, KL, E, dt, dev), True, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("pad", ["lens", "random", "ref"])
@pytest.mark.parametrize("causal", [False, True])
def test_padmask(pkg, dev, tune, dt, E, pad, causal):
    if dt == "f16" and causal:
        pytest.skip("covered in bf16")
    tune(bwd_w64=1)
    check(pkg, make_inputs(94, 3, 2, 2, 700, 700, E, dt, dev, pad=pad), causal, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("QH,KH", [(4, 1), (6, 2), (8, 2)])
@pytest.mark.parametrize("causal", [False, True])
def test_gqa(pkg, dev, tune, dt, E, QH, KH, causal):
    """the dK/dV pass sweeps the q-heads of a kv head in one stream (ragged QL: every head's last step runs into the next head's
    rows, which the padded row constants turn into P = 0)"""
    if dt == "f16" and (QH, KH) != (8, 2):
        pytest.skip("covered in bf16")
    tune(bwd_w64=1)
    check(pkg, make_inputs(95, 2, QH, KH, 515, 515, E, dt, dev), causal, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
def test_fully_masked_batch_and_dead_rows(pkg, dev, tune, dt, E):
    """a batch without any valid key (dK = dV = 0 there, dQ = 0) and, under the causal mask with left padding, query rows that see
    no key: zero dQ rows, finite dK / dV"""
    tune(bwd_w64=1)
    d = make_inputs(96, 3, 2, 2, 300, 300, E, dt, dev)
    m = np.ones((3, 300), dtype=bool)
    m[1, :] = False
    m[2, :70] = False                    # left padding: causal queries 0..69 of batch 2 see nothing
    d["mask"] = torch.tensor(m).to(dev)
    for causal in (False, True):
        dq, dk, dv = run_bwd(pkg, d, causal)
        assert torch.isfinite(dk.float()).all() and torch.isfinite(dv.float()).all() and torch.isfinite(dq.float()).all()
        assert float(dk[1].float().abs().max()) == 0.0 and float(dv[1].float().abs().max()) == 0.0
        assert float(dq[1].float().abs().max()) == 0.0
        if causal:
            assert float(dq[2, :, :70].float().abs().max()) == 0.0
        assert float(dk[2, :, :70].float().abs().max()) == 0.0
        # the live part against the oracle (dead rows give NaN in the naive formula: compare batch 0 and the live rows of batch 2)
        rq, rk, rv, _ = oracle_bwd({**d, "mask": d["mask"]}, causal)
        assert_close("dk0", dk[0], rk[0], dt, kind="grad")
        assert_close("dv0", dv[0], rv[0], dt, kind="grad")
        assert_close("dq0", dq[0], rq[0], dt, kind="grad")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E,causal,pad,QH,KH", [(64, False, None, 2, 2), (64, True, "ref", 4, 2), (128, False, None, 2, 2), (128, True, "lens", 4, 2)])
def test_bitwise_reproducible_and_close_to_the_32_row_form(pkg, dev, tune, dt, E, causal, pad, QH, KH):
    d = make_inputs(97, 2, QH, KH, 1100, 1100, E, dt, dev, pad=pad)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])
    flush = torch.empty(288 * 1024 * 1024, dtype=torch.uint8, device=dev)      # > the 256 MiB Infinity Cache

    def bwd():
        g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        return g[:3]

    tune(bwd_w64=1)
    outs = []
    for _ in range(4):
        flush.fill_(1)
        outs.append(bwd())
    for other in outs[1:]:
        for a, b, name in zip(outs[0], other, ("dq", "dk", "dv")):
            assert torch.equal(a, b), name
    tune(bwd_w64=0)
    ref = bwd()
    # same residuals, same products, another summation order and fp32 exponent arithmetic of another shape: equal to rounding
    for a, b, name in zip(outs[0], ref, ("dq", "dk", "dv")):
        scale = float(b.float().abs().max())
        assert float((a.float() - b.float()).abs().max()) <= (1.6e-2 if dt == "bf16" else 2e-3) * scale, name


@pytest.mark.parametrize("E", [64, 128])
def test_long_sweep(pkg, dev, tune, E):
    """many iterations of the ring (L = 2304: 72 / 36 steps per workgroup; the full-size launches of test_baseline_configs_gpu.py
    go to 512 steps)"""
    tune(bwd_w64=1)
    d = make_inputs(98, 1, 2, 1, 2304, 2304, E, "bf16", dev)
    check(pkg, d, False, "bf16")
    check(pkg, d, True, "bf16")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("QL,KL,QH,KH,causal,pad", [(256, 256, 2, 2, False, None), (255, 257, 2, 2, False, None), (700, 300, 4, 2, True, "ref"),
                                                    (515, 515, 6, 2, False, "lens"), (1024, 1024, 2, 1, True, None), (33, 1000, 2, 2, False, "random")])
def test_e256(pkg, dev, tune, dt, QL, KL, QH, KH, causal, pad):
    """E = 256 (16-bit): shapes 1 x 1 -- one 32-row block each way -- and the dK/dV pass split into two column halves per key block
    (every workgroup keeps full-E K / V fragments and half of the accumulators); spill-free where the 32-row kernels of fa_bwd.hpp
    spilled 126..589 registers at the 256-register cap."""
    tune(bwd_w64=1)
    check(pkg, make_inputs(99, 2, QH, KH, QL, KL, 256, dt, dev, pad=pad), causal, dt)


@pytest.mark.parametrize("E", [64, 128])
def test_misaligned_bases_are_refused_through_the_c_abi(pkg, dev, E):
    """NNOP_ERR_ALIGN (include/nnop_hip.h, round 4): the kernels move tensors and the workspace with 16-byte vector accesses and LDS-DMA
    from the raw base, so an offset pointer is refused by the boundary BEFORE anything is launched -- forward (q 8 bytes off) and backward
    (workspace 8 bytes off; round 3 fell back to the 32-row dK/dV kernel instead) -- and the aligned call right behind it still works."""
    d = make_inputs(90, 2, 4, 2, 300, 300, E, "bf16", dev)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], None, causal=True, kpad_mask=None)
    # forward: a q that starts 8 bytes into its allocation
    raw_q = torch.empty(d["q"].numel() + 8, dtype=d["q"].dtype, device=dev)
    q_off = raw_q[4:4 + d["q"].numel()].view_as(d["q"])
    q_off.copy_(d["q"])
    assert q_off.data_ptr() % 16 == 8
    o2, ms2, ls2 = torch.empty_like(o), torch.empty_like(ms), torch.empty_like(ls)
    with pytest.raises(pkg.NNopError) as e:
        pkg.fa_fwd_into(o2, ms2, ls2, q_off, d["k"], d["v"], causal=True)
    assert e.value.status == pkg._lib.NNOP_ERR_ALIGN
    # backward: the workspace 8 bytes off
    nbytes = pkg.bwd_workspace_bytes(d["q"], d["k"], d["v"], causal=True)
    raw = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
    dq, dk, dv = torch.empty_like(d["q"]), torch.empty_like(d["k"]), torch.empty_like(d["v"])
    with pytest.raises(pkg.NNopError) as e:
        pkg.fa_bwd_into(dq, dk, dv, None, raw[8:8 + nbytes], d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=True)
    assert e.value.status == pkg._lib.NNOP_ERR_ALIGN
    pkg.fa_bwd_into(dq, dk, dv, None, raw[:nbytes], d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=True)
    torch.cuda.synchronize()
    rq, rk, rv, _ = oracle_bwd(d, True)
    assert_close("dq", dq, rq, "bf16", kind="grad")
    assert_close("dk", dk, rk, "bf16", kind="grad")
    assert_close("dv", dv, rv, "bf16", kind="grad")


@pytest.mark.parametrize("dt,E,QH,KH,L,causal,pad,ragged", [
    ("bf16", 64, 16, 16, 2048, True, None, 0), ("bf16", 64, 16, 16, 2048, False, "lens", 0), ("f16", 128, 16, 16, 2048, True, "lens", 13),
    ("f16", 128, 16, 16, 2048, False, None, 37), ("bf16", 128, 16, 8, 4096, True, None, 0),
    ("bf16", 256, 16, 16, 2048, True, "lens", 0)])         # E = 256: the dQ pass only (its dK/dV pass runs every block twice, column halves)
def test_persistent_block_list_is_bitwise_the_one_block_per_workgroup_launch(pkg, dev, tune, dt, E, QH, KH, L, causal, pad, ragged):
    """The persistent form of the backward kernels (256 workgroups walking a static balanced block list, csrc/fa_bwd_w64.hpp, knob
    bwd_persist) runs the same per-block code as the launch with one workgroup per block: dq, dk, dv are bitwise equal -- every
    block visited exactly once, for both passes (dQ: columns = batch x q-head; dK/dV: batch x kv-head, the q-heads of a group
    streamed inside a block) -- with ragged lengths and key padding; repeated launches bitwise equal (the hand-over between two
    blocks is one barrier: the ring, the V image and the validity words are rewritten behind it)."""
    B = 4
    d = make_inputs(79, B, QH, KH, L - ragged, L - (2 * ragged if ragged else 0), E, dt, dev, pad=pad)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)

    def bwd():
        g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], None, causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        return g[:3]

    tune(bwd_w64=1, bwd_persist=0)
    ref = bwd()
    tune(bwd_w64=1, bwd_persist=1)
    for _ in range(3):
        flush.fill_(1)
        got = bwd()
        for a, b_, name in zip(ref, got, ("dq", "dk", "dv")):
            assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b_.float())), name


def test_narrow_shape_follows_the_grid(pkg, dev, tune):
    """the launcher's own choice (knob at automatic) is bitwise the forced shape its rule names: 256 CUs -- 32 blocks of 256 rows leave
    7/8 of the chip idle -> 32-row waves; 512 blocks -> 64-row waves; causal: narrow up to a round and a half of 256-row blocks"""
    for B, causal, want in ((1, False, 1), (16, False, 0), (8, True, 1), (16, True, 0)):
        d = make_inputs(97, B, 8, 8, 1024, 1024, 64, "bf16", dev)
        tune(bwd_narrow=-1)
        a = run_bwd(pkg, d, causal)
        tune(bwd_narrow=want)
        b = run_bwd(pkg, d, causal)
        assert all(torch.equal(x, y) for x, y in zip(a, b)), (B, causal)
