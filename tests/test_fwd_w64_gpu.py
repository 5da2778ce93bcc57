"""GPU parity of the 64-rows-per-wave forward (csrc/fa_fwd_w64.hpp) -- forced on with the test hook for BOTH of its
instantiations (the launcher itself picks it only where it measured faster: E = 128 on large grids) -- against the fp64
oracle, in every mode it serves: plain, causal, key padding (reference pattern, prefix lengths, random), ragged QL / KL, GQA,
tile counts 1, 2, 3 (ring start-up), odd and even; plus what a tolerance check can miss in a hand-placed, inline-asm kernel:
bitwise reproducibility across launches, equality of the residual contract with the 32-row forms, and a forced rescale of
the deferred-max reference at chosen tiles (the rare branch that re-bases O, l and -- with the scale folded into Q -- the
pending score tile)."""
import numpy as np
import pytest
import torch

from util import RTOL, assert_close, make_inputs, oracle_fwd

pytestmark = pytest.mark.gpu


def run(pkg, d, causal):
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
    torch.cuda.synchronize()
    return o, ms, ls


def check(pkg, d, causal, dt):
    o, ms, ls = run(pkg, d, causal)
    o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
    assert_close("o", o, o_ref, dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        lse = ms.double().cpu().numpy() + np.log(ls.double().cpu().numpy())
        lse_ref = ms_ref + np.log(ls_ref)
    assert_close("lse", lse, lse_ref, dt)
    assert_close("ms", ms, ms_ref, dt)
    return o, ms, ls


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("QL,KL", [(256, 64), (300, 128), (64, 192), (511, 256), (512, 1024), (1024, 320), (257, 704)])
def test_plain(pkg, dev, tune, dt, E, QL, KL):
    tune(fwd_w64=1)
    check(pkg, make_inputs(71, 2, 2, 2, QL, KL, E, dt, dev, need_do=False), False, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("L", [63, 64, 255, 256, 257, 511, 777, 1024])
@pytest.mark.parametrize("pad", [None, "ref"])
def test_causal(pkg, dev, tune, dt, E, L, pad):
    if pad == "ref" and L < 64:
        pytest.skip("the reference pattern masks the last 11 keys")
    tune(fwd_w64=1)
    check(pkg, make_inputs(72, 2, 2, 2, L, L, E, dt, dev, pad=pad, need_do=False), True, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("pad", ["ref", "lens", "random"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("QL,KL", [(700, 700), (300, 1000), (512, 448)])
def test_key_padding_and_ragged(pkg, dev, tune, dt, E, pad, causal, QL, KL):
    tune(fwd_w64=1)
    check(pkg, make_inputs(73, 3, 2, 2, QL, KL, E, dt, dev, pad=pad, need_do=False), causal, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("QH,KH", [(4, 1), (6, 2), (8, 2)])
@pytest.mark.parametrize("causal", [False, True])
def test_gqa(pkg, dev, tune, dt, QH, KH, causal):
    tune(fwd_w64=1)
    check(pkg, make_inputs(74, 2, QH, KH, 515, 515, 128, dt, dev, need_do=False), causal, dt)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E,causal,pad", [(64, False, None), (64, True, "ref"), (128, False, None), (128, True, "lens")])
def test_bitwise_reproducible_and_same_residuals_as_the_32_row_form(pkg, dev, tune, dt, E, causal, pad):
    """Repeated launches (caches flushed in between) are bitwise identical -- the screen for races of the LDS-DMA ring and
    for reads of an accumulator tile before its MFMA has landed (both would come and go with timing); `ms`, `o` agree with the
    32-row form to rounding (scale folded into Q, different summation order of the keys)."""
    d = make_inputs(75, 2, 4, 2, 1100, 1100, E, dt, dev, pad=pad, need_do=False)
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
    tune(fwd_w64=1)
    outs = []
    for _ in range(4):
        flush.fill_(1)
        outs.append(run(pkg, d, causal))
    for other in outs[1:]:
        for a, b, name in zip(outs[0], other, ("o", "ms", "ls")):
            assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), name
    tune(fwd_w64=0)
    o0, ms0, ls0 = run(pkg, d, causal)
    # ms: the row max of the scaled logits.  The 64-row form folds scale * log2(e) into Q (rounded to T once), so its max is
    # the max of slightly different logits: equal to the 32-row form's to the rounding of T (one unit in the last place of
    # ms itself plus the 2^-9 / 2^-12 relative rounding of the folded Q), not bitwise.
    m1, m0 = torch.nan_to_num(outs[0][1].float()), torch.nan_to_num(ms0.float())
    assert float((m1 - m0).abs().max()) <= (1.6e-2 if dt == "bf16" else 2e-3) * float(m0.abs().max())
    scale = float(torch.nan_to_num(o0.float()).abs().max())
    assert float(torch.nan_to_num(outs[0][0].float() - o0.float()).abs().max()) <= (1.5e-2 if dt == "bf16" else 2e-3) * scale


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("exact", [0, 1])
@pytest.mark.parametrize("spike_tiles", [(1,), (2, 5), (0, 3, 4, 9)])
def test_forced_rescale_of_the_deferred_max(pkg, dev, tune, dt, E, exact, spike_tiles):
    """cdna_hip_programming.md rule 26: the rescale branch fires only when a row's max outgrows the exponent reference by 2^8
    -- never on N(0,1) data after the first tile.  Plant keys that are strongly aligned with some queries at chosen kv
    tiles so that the running max jumps by far more than the threshold there, for a subset of the rows of a wave (both
    query blocks, not all lanes), and compare the FULL output with the oracle.

    Both scale forms: `exact` (fp32 scale inside the exponent) must meet the standard tolerance at any logit size; the
    default form folds scale * log2(e) into Q, so a logit x carries a relative rounding of 2^-9 (bf16) / 2^-12 (fp16): planted
    logits of |x| ~ 50 .. 200 (log2 units) are then off by up to |x| 2^-9 in the exponent, and the weights of near-tied top
    keys by that factor -- the error model the tolerance below states (it is what the reference's own `S * scale` in T does)."""
    rng = np.random.default_rng(76)
    B, H, L = 1, 2, 704
    d = make_inputs(77, B, H, H, L, L, E, dt, dev, need_do=False)
    q, k = d["q"].float().cpu().numpy(), d["k"].float().cpu().numpy()
    for i, t in enumerate(spike_tiles):
        rows = rng.choice(L, size=40, replace=False)                 # queries that will see the spike
        key = 64 * t + int(rng.integers(0, 64))
        direction = rng.standard_normal(E).astype(np.float32)
        direction /= np.linalg.norm(direction)
        gain = 6.0 * (i + 1) * np.sqrt(E)
        k[:, :, key] = direction * gain
        q[:, :, rows] = q[:, :, rows] * 0.2 + direction * 6.0
    tdt = d["q"].dtype
    d["q"], d["k"] = torch.tensor(q).to(tdt).to(dev), torch.tensor(k).to(tdt).to(dev)
    tune(fwd_w64=1, fwd_exact_scale=exact)
    if exact:
        check(pkg, d, False, dt)
        check(pkg, d, True, dt)
        return
    # folded scale: exponent error <= max|logit| * eps(T) -> relative weight error of tied keys; o is a convex combination
    # of |v| <= ~4.5, so the element-wise bound is that factor times the value range
    xmax = 36.0 * len(spike_tiles) * 1.4427                          # largest planted logit, log2 units
    eps = 2.0 ** -9 if dt == "bf16" else 2.0 ** -12
    bound = (2.0 ** (xmax * eps) - 1.0) * 4.5 + (2e-2 if dt == "bf16" else 3e-3)
    for causal in (False, True):
        o, ms, ls = run(pkg, d, causal)
        o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
        err = np.abs(np.nan_to_num(o.double().cpu().numpy()) - np.nan_to_num(o_ref))
        assert err.max() <= bound, f"o: max err {err.max():.3e} > error-model bound {bound:.3e}"
        assert_close("ms", ms, ms_ref, dt, scale=1.0 + xmax * eps / RTOL[dt])


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
def test_scale_fold_crossover_is_why_the_exact_scale_is_the_default(pkg, dev, tune, dt, E):
    """Where does folding scale * log2(e) into Q (rounded to T once; the opt-in NNOP_FWD_EXACT_SCALE=0, 8-12 % faster) leave the
    standard parity tolerance?  Two near-tied dominant keys per query with ORTHOGONAL supports (the rounding of the folded Q is per
    query channel: keys that share their direction see the same error and it cancels in the softmax), logits |s * scale| = M built
    dense (spread over all channels) or sparse (one outlier channel per key -- the massive-activation pattern of trained models).
    The DEFAULT path (exact fp32 scale) must hold the standard `assert_close` tolerance over the whole sweep; the folded form is
    measured beside it: fine on small logits, outside the tolerance at |s * scale| = 30 with sparse keys -- inside what a trained
    model produces, which is why it is not the default (round-2 verdict item 9; numbers: profiles/r03/fold_sweep.log)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fold_sweep import planted
    from util import ATOL_FRAC
    ratios = {}
    for structure in ("dense", "sparse"):
        for M in (2, 12, 30, 60):
            q, k, v = (x.to(dev) for x in planted(M, E, 1024, structure, dt))
            d = dict(q=q, k=k, v=v, pair=None, mask=None)
            o_ref, _, _ = oracle_fwd(d, False)
            tol = ATOL_FRAC[dt] * np.abs(o_ref).max() + RTOL[dt] * np.abs(o_ref)
            for exact in (1, 0):
                tune(fwd_w64=1, fwd_exact_scale=exact)
                o = run(pkg, d, False)[0]
                ratios[(structure, M, exact)] = float((np.abs(o.double().cpu().numpy() - o_ref) / tol).max())
            tune(fwd_w64=1, fwd_exact_scale=-1)                       # the default
            check(pkg, d, False, dt)
    assert max(r for (s, M, ex), r in ratios.items() if ex == 1) <= 0.75, ratios
    assert ratios[("dense", 2, 0)] <= 1.0 and ratios[("sparse", 2, 0)] <= 1.0, ratios          # the fold is fine on small logits
    assert max(ratios[("sparse", 30, 0)], ratios[("sparse", 60, 0)]) > 1.0, ratios            # ... and is not on large sparse ones


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E", [64, 128])
@pytest.mark.parametrize("QL,KL,causal,pad", [(512, 1024, False, None), (777, 777, True, "ref"), (700, 700, True, "lens"), (300, 1000, False, "random")])
def test_folded_scale_variant_in_every_mode(pkg, dev, tune, dt, E, QL, KL, causal, pad):
    """the opt-in form (scale folded into Q; PRE = true instantiations) keeps its own coverage of the modes now that the exact
    scale is the default: N(0,1) logits are small, where the fold holds the standard tolerance"""
    tune(fwd_w64=1, fwd_exact_scale=0)
    check(pkg, make_inputs(78, 2, 4, 2, QL, KL, E, dt, dev, pad=pad, need_do=False), causal, dt)


@pytest.mark.parametrize("dt,E,causal,pad,QL,KL", [
    ("bf16", 64, True, None, 2048, 2048), ("bf16", 64, False, "random", 2048, 2048), ("f16", 128, True, "lens", 2048 - 13, 2048 - 37),
    ("f16", 128, False, None, 2048, 1024 + 37), ("bf16", 128, True, None, 2048, 2048), ("bf16", 128, False, "lens", 2048, 2048)])
def test_persistent_block_list_is_bitwise_the_one_block_per_workgroup_launch(pkg, dev, tune, dt, E, causal, pad, QL, KL):
    """The persistent form of the 64-row forward (256 workgroups walking a static, balanced block list; csrc/fa_fwd_w64.hpp,
    knob fwd_persist) runs the same per-block code as the launch with one workgroup per block: outputs and residuals are bitwise
    equal, every block is visited exactly once (a block missed or done twice by the list shows as a difference), also with ragged
    lengths, key padding and GQA -- and repeated persistent launches are bitwise equal among themselves (the hand-over between two
    blocks of a workgroup is one barrier: a ring slot overwritten too early would come and go with timing).  One slice against the oracle."""
    d = make_inputs(77, 4, 16, 4, QL, KL, E, dt, dev, pad=pad, need_do=False)        # B x QH = 64 columns: 8 per XCD, 2 steps of 32 blocks
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
    tune(fwd_w64=1, fwd_persist=0)
    ref = run(pkg, d, causal)
    tune(fwd_w64=1, fwd_persist=1)
    for _ in range(3):
        flush.fill_(1)
        got = run(pkg, d, causal)
        for a, b, name in zip(ref, got, ("o", "ms", "ls")):
            assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), name
    from oracle.naive_attention import naive_attention_slice
    b, kh = 3, 2
    hs = slice(kh * 4, kh * 4 + 4)
    f64 = lambda t: t.double().cpu().numpy()
    o_ref = naive_attention_slice(f64(d["q"][b, hs]), f64(d["k"][b, kh]), f64(d["v"][b, kh]), None, causal=causal,
                                  kpad_mask=None if d["mask"] is None else d["mask"][b].cpu().numpy())["o"]
    from util import assert_close
    assert_close("o", got[0][b, hs], o_ref, dt)
