"""Seeded randomized parity sweep for flash attention through the C ABI: random shapes (ragged query / key counts, 1..5
batches, GQA ratios), E, dtype, causal, key-padding kind and pair bias, forward and backward against the fp64 oracle.
Complements the structured grids (test_fwd_gpu.py, test_bwd_gpu.py): same checker, same tolerances (tests/util.py), the
shapes are not hand-picked.  Deterministic: the case list is a pure function of CASE_SEED."""
import os

import numpy as np
import pytest
import torch

from util import assert_close, make_inputs, oracle_bwd, oracle_fwd

pytestmark = pytest.mark.gpu
# (NNOP_FUZZ_SEED: another deterministic case list -- for longer offline sweeps; the suite runs the default)
CASE_SEED, N_CASES = int(os.environ.get("NNOP_FUZZ_SEED", "20251")), 48


def _cases():
    rng = np.random.default_rng(CASE_SEED)
    out = []
    for i in range(N_CASES):
        E = int(rng.choice([16, 32, 64, 128]))
        dt = str(rng.choice(["f32", "bf16", "f16"]))
        KH = int(rng.choice([1, 2, 3]))
        QH = KH * int(rng.choice([1, 1, 2, 4]))
        B = int(rng.integers(1, 4))
        QL = int(rng.integers(1, 420))
        KL = int(rng.integers(1, 420)) if rng.random() < 0.6 else QL
        causal = bool(rng.random() < 0.5)
        pad = [None, None, "lens", "random"][int(rng.integers(0, 4))]
        pair = bool(rng.random() < 0.25) and E <= 64
        out.append((i, dt, E, B, QH, KH, QL, KL, causal, pad, pair))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "{}-{}-E{}-B{}-H{}x{}-L{}x{}-c{}-{}-p{}".format(*c[:9], c[9] or "nopad", int(c[10])))
def test_random_case(pkg, dev, case):
    i, dt, E, B, QH, KH, QL, KL, causal, pad, pair = case
    d = make_inputs(1000 + i, B, QH, KH, QL, KL, E, dt, dev, pair=pair, pad=pad)
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], d["pair"], causal=causal, kpad_mask=d["mask"])
    dq, dk, dv, dp = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], d["pair"], causal=causal,
                                              kpad_mask=d["mask"])
    torch.cuda.synchronize()
    o_ref, ms_ref, ls_ref = oracle_fwd(d, causal)
    # rows that see no key (causal with QL > KL cannot happen top-left aligned; padding can hide every key) are NaN
    # in the oracle and in the kernel alike -- assert_close compares the NaN pattern
    assert_close("o", o, o_ref, dt, floor=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        lse_ref = ms_ref + np.log(ls_ref)
        lse = ms.double().cpu().numpy() + np.log(ls.double().cpu().numpy())
    fin = np.isfinite(lse_ref)
    assert (np.isfinite(lse) == fin).all()
    assert_close("lse", np.where(fin, lse, 0.0), np.where(fin, lse_ref, 0.0), dt, floor=True)
    rq, rk, rv, rp = oracle_bwd(d, causal)
    sc = 1.0 if dt == "f32" else 2.0
    dead = np.isnan(o_ref).any(axis=-1)                       # rows with no visible key
    if dead.any():
        # deviation 3 of DESIGN.md section 2: such rows get dq = 0 and contribute nothing to dk / dv, where the naive
        # formula poisons the whole (batch, kv-head) with NaN -- compare on the oracle evaluated without them
        assert torch.isfinite(dq).all() and torch.isfinite(dk).all() and torch.isfinite(dv).all()
        assert (dq.double().cpu().numpy()[dead] == 0).all()
        return
    assert_close("dq", dq, rq, dt, sc, floor=True)
    assert_close("dk", dk, rk, dt, sc, floor=True)
    assert_close("dv", dv, rv, dt, sc, floor=True)
    if pair:
        assert_close("dpair", dp, rp, dt, sc, floor=True)


def _cases_w64():
    """second sweep for the kernels the benchmarks run: 16-bit, E in {64, 128}, lengths up to 1500 (several workgroups, many kv
    tiles), no pair bias, the 64-row forward forced for half of the cases (the backward's one-wave-per-SIMD form is the default)"""
    rng = np.random.default_rng(CASE_SEED + 7)
    out = []
    for i in range(28):
        E = int(rng.choice([64, 128]))
        dt = str(rng.choice(["bf16", "f16"]))
        KH = int(rng.choice([1, 2]))
        QH = KH * int(rng.choice([1, 2, 4]))
        B = int(rng.integers(1, 3))
        QL = int(rng.integers(1, 1500))
        KL = int(rng.integers(1, 1500)) if rng.random() < 0.5 else QL
        causal = bool(rng.random() < 0.5)
        pad = [None, None, "lens", "random"][int(rng.integers(0, 4))]
        w64 = int(rng.choice([-1, 1]))
        out.append((i, dt, E, B, QH, KH, QL, KL, causal, pad, w64))
    return out


def _narrow_of(i):
    """backward wave shape of case i of the second sweep: the launcher's choice / 64 stationary rows / 32 (BwdW64Shape NARROW)"""
    return (-1, 0, 1)[i % 3]


@pytest.mark.parametrize("case", _cases_w64(), ids=lambda c: "{}-{}-E{}-B{}-H{}x{}-L{}x{}-c{}-{}-w{}".format(*c[:9], c[9] or "nopad", c[10]))
def test_random_case_on_the_64_row_kernels(pkg, dev, tune, case):
    i, dt, E, B, QH, KH, QL, KL, causal, pad, w64 = case
    tune(fwd_w64=w64, bwd_narrow=_narrow_of(i))
    test_random_case(pkg, dev, (3000 + i, dt, E, B, QH, KH, QL, KL, causal, pad, False))


def _row_cases():
    rng = np.random.default_rng(CASE_SEED + 1)
    out = []
    for i in range(40):
        dt = str(rng.choice(["f32", "bf16", "f16"]))
        wdt = "f32" if (dt == "f32" or rng.random() < 0.5) else dt
        n = int(rng.integers(1, 70))
        emb = int(rng.choice([int(rng.integers(1, 300)), 8 * int(rng.integers(1, 1200)), 4 * int(rng.integers(1, 3000)),
                              int(rng.integers(300, 20000))]))
        out.append((i, dt, wdt, n, emb))
    return out


def _cases_duo():
    """third sweep: the two-waves-per-SIMD forward (csrc/fa_fwd_duo.hpp) forced on random 16-bit shapes -- any length from 1, ragged,
    GQA, every mask kind -- so that odd tile counts, key groups without a tile and waves without rows all occur; its three generated
    loops in turn: E = 64 with 64-row waves (knob 2), with 32-row waves (3), E = 128 (1)"""
    rng = np.random.default_rng(CASE_SEED + 11)
    out = []
    for i in range(24):
        dt = str(rng.choice(["bf16", "f16"]))
        KH = int(rng.choice([1, 2]))
        QH = KH * int(rng.choice([1, 2, 4]))
        B = int(rng.integers(1, 3))
        QL = int(rng.integers(1, 1500))
        KL = int(rng.integers(1, 1500)) if rng.random() < 0.5 else QL
        causal = bool(rng.random() < 0.5)
        pad = [None, None, "lens", "random"][int(rng.integers(0, 4))]
        out.append((i, dt, (64, 64, 128)[i % 3], B, QH, KH, QL, KL, causal, pad))
    return out


@pytest.mark.parametrize("case", _cases_duo(), ids=lambda c: "{}-{}-E{}-B{}-H{}x{}-L{}x{}-c{}-{}".format(*c[:9], c[9] or "nopad"))
def test_random_case_on_the_two_wave_forward(pkg, dev, tune, case):
    i, dt, E, B, QH, KH, QL, KL, causal, pad = case
    tune(fwd_duo=1 if E == 128 else 2 + i % 3)
    test_random_case(pkg, dev, (4000 + i, dt, E, B, QH, KH, QL, KL, causal, pad, False))


@pytest.mark.parametrize("case", _row_cases(), ids=lambda c: "{}-{}-w{}-n{}-emb{}".format(*c))
def test_random_row_ops(pkg, dev, case):
    """Softmax, RMSNorm, LayerNorm (forward + pullback) on random row counts / lengths: every register shape, the
    generic kernels for odd lengths, rows longer than any register shape."""
    from oracle.naive_norms import naive_layer_norm, naive_layer_norm_grads, naive_rms_norm, naive_rms_norm_grads
    from oracle.naive_softmax import naive_softmax, naive_softmax_grad
    from util import TORCH_DT
    i, dt, wdt, n, emb = case
    rng = np.random.default_rng(5000 + i)
    t = lambda a, d_: torch.tensor(np.asarray(a, np.float32)).to(TORCH_DT[d_]).to(dev)
    x, dy = t(rng.standard_normal((n, emb)) * 1.5, dt), t(rng.standard_normal((n, emb)), dt)
    w, b = t(rng.standard_normal(emb), wdt), t(rng.standard_normal(emb), wdt)
    f64 = lambda z: z.detach().to(torch.float64).cpu().numpy()
    rt = {"f32": 1e-5, "f16": 2e-3, "bf16": 1.6e-2}[dt]

    def close(got, ref, k=1.0, extra=0.0):
        np.testing.assert_allclose(f64(got), ref, rtol=rt * k, atol=rt * k * max(np.abs(ref).max(), 1e-30) + extra)

    y = pkg.online_softmax(x)
    close(y, naive_softmax(f64(x)), extra=1e-9)
    close(pkg.grad_online_softmax(dy, y), naive_softmax_grad(f64(dy), f64(y)), k=2.0)
    yr, rms = pkg._rms_norm(x, w, offset=0.25, eps=1e-5)
    close(yr, naive_rms_norm(f64(x), f64(w), offset=0.25, eps=1e-5)[0])
    dx, dw = pkg.grad_rms_norm(dy, rms, x, w, offset=0.25)
    rdx, rdw = naive_rms_norm_grads(f64(dy), f64(x), f64(w), offset=0.25, eps=1e-5)
    close(dx, rdx, k=2.0)
    np.testing.assert_allclose(f64(dw), rdw, rtol=2e-4, atol=2e-5 * max(np.abs(rdw).max(), 1.0) * max(1.0, n / 8))
    yl, mu, sg = pkg._layer_norm(x, w, b, eps=1e-5)
    close(yl, naive_layer_norm(f64(x), f64(w), f64(b), eps=1e-5)[0], k=2.0)
    if emb > 1:      # emb == 1: variance 0, the pullback is 0 * rstd ~ 1e5-amplified rounding noise in any implementation
        dx, dw, db = pkg.grad_layer_norm(dy, mu, sg, x, w, b)
        rdx, rdw, rdb = naive_layer_norm_grads(f64(dy), f64(x), f64(w), eps=1e-5)
        close(dx, rdx, k=3.0)
        wt = {"f32": 2e-4, "f16": 2e-3, "bf16": 1.6e-2}[wdt]
        np.testing.assert_allclose(f64(dw), rdw, rtol=wt, atol=wt * max(np.abs(rdw).max(), 1.0) * max(1.0, n / 8) * 0.2)
        np.testing.assert_allclose(f64(db), rdb, rtol=wt, atol=wt * max(np.abs(rdb).max(), 1.0) * max(1.0, n / 8) * 0.2)
