"""CPU tests of the oracle itself (no GPU): pins the numpy restatement against
(1) an independent implementation (torch CPU fp64 softmax(QK^T)V + autograd),
(2) its own tiled restatement of the reference's recurrences (src/attention.jl:44-130,
    src/attention_bwd.jl:39-197) including the (ms, ls) residual contract,
(3) the committed golden fixtures, and (4) the work model of SURVEY.md section 8(d)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.naive_attention import (attention_bytes, attention_flops, naive_attention,
                                    naive_attention_f32, naive_attention_f32_fwd_bwd,
                                    naive_attention_grads, tiled_flash_bwd, tiled_flash_fwd)

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fa_*.npz")))


def rand_case(seed, B, QH, KH, QL, KL, E, pair=False, pad=None):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((B, QH, QL, E))
    k = rng.standard_normal((B, KH, KL, E))
    v = rng.standard_normal((B, KH, KL, E))
    do = rng.standard_normal((B, QH, QL, E))
    p = rng.standard_normal((B, KL, QL, QH)) if pair else None
    m = None
    if pad == "ref":
        m = np.ones((B, KL), bool)
        m[-1, -11:] = False
    elif pad == "lens":
        m = np.arange(KL)[None, :] < rng.integers(1, KL + 1, size=B)[:, None]
    return q, k, v, do, p, m


def torch_attention(q, k, v, do, pair, causal, mask):
    """Independent implementation: torch ops + autograd, fp64, CPU."""
    tq, tk, tv = (torch.tensor(x, dtype=torch.float64, requires_grad=True) for x in (q, k, v))
    tp = torch.tensor(pair, dtype=torch.float64, requires_grad=True) if pair is not None else None
    rep = q.shape[1] // k.shape[1]
    ke, ve = tk.repeat_interleave(rep, dim=1), tv.repeat_interleave(rep, dim=1)
    a = (tq @ ke.transpose(-1, -2)) / (q.shape[-1] ** 0.5)
    if tp is not None:
        a = a + tp.permute(0, 3, 2, 1)
    QL, KL = q.shape[2], k.shape[2]
    if causal:
        keep = torch.arange(KL)[None, :] <= torch.arange(QL)[:, None]
        a = a.masked_fill(~keep, float("-inf"))
    if mask is not None:
        a = a.masked_fill(~torch.tensor(mask)[:, None, None, :], float("-inf"))
    o = torch.softmax(a, dim=-1) @ ve
    o.backward(torch.tensor(do, dtype=torch.float64))
    g = lambda t: None if t is None else t.grad.numpy()
    return o.detach().numpy(), g(tq), g(tk), g(tv), g(tp)


CASES = [
    (0, 2, 2, 2, 50, 50, 16, False, False, None),
    (1, 2, 2, 2, 33, 70, 32, False, True, "ref"),
    (2, 2, 2, 2, 64, 64, 64, True, False, None),
    (3, 2, 6, 2, 40, 40, 32, True, False, None),
    (4, 2, 4, 1, 31, 31, 16, True, True, "ref"),
    (5, 3, 2, 2, 20, 90, 128, False, False, "lens"),
]


@pytest.mark.parametrize("seed,B,QH,KH,QL,KL,E,causal,pair,pad", CASES)
def test_oracle_vs_torch_autograd(seed, B, QH, KH, QL, KL, E, causal, pair, pad):
    q, k, v, do, p, m = rand_case(seed, B, QH, KH, QL, KL, E, pair, pad)
    o = naive_attention(q, k, v, p, causal=causal, kpad_mask=m)
    dq, dk, dv, dp = naive_attention_grads(q, k, v, do, p, causal=causal, kpad_mask=m)
    o_t, dq_t, dk_t, dv_t, dp_t = torch_attention(q, k, v, do, p, causal, m)
    np.testing.assert_allclose(o, o_t, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dq, dq_t, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dk, dk_t, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dv, dv_t, rtol=1e-9, atol=1e-11)
    if pair:
        np.testing.assert_allclose(dp, dp_t, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("seed,B,QH,KH,QL,KL,E,causal,pair,pad", CASES)
@pytest.mark.parametrize("gsz", [16, 64])
def test_tiled_restatement_matches_naive(seed, B, QH, KH, QL, KL, E, causal, pair, pad, gsz):
    q, k, v, do, p, m = rand_case(seed, B, QH, KH, QL, KL, E, pair, pad)
    o, ms, ls = naive_attention(q, k, v, p, causal=causal, kpad_mask=m, return_stats=True)
    o_t, ms_t, ls_t = tiled_flash_fwd(q, k, v, p, causal=causal, kpad_mask=m, gsz=gsz)
    np.testing.assert_allclose(o_t, o, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(ms_t, ms, rtol=1e-12, atol=1e-13)   # ms is the row max
    np.testing.assert_allclose(ls_t, ls, rtol=1e-12)
    dq, dk, dv, dp = naive_attention_grads(q, k, v, do, p, causal=causal, kpad_mask=m)
    dq_t, dk_t, dv_t, dp_t = tiled_flash_bwd(do, o_t, ms_t, ls_t, q, k, v, p, causal=causal, kpad_mask=m, gsz=gsz)
    np.testing.assert_allclose(dq_t, dq, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dk_t, dk, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(dv_t, dv, rtol=1e-9, atol=1e-11)
    if pair:
        np.testing.assert_allclose(dp_t, dp, rtol=1e-9, atol=1e-11)


def test_fully_masked_tile_is_finite_and_dead_rows_are_nan():
    """a fully masked K tile must not poison the result (the naive formula
    is finite there); a row with NO visible key is 0/0 = NaN in the naive formula."""
    q, k, v, do, _, _ = rand_case(9, 2, 2, 2, 16, 96, 16)
    m = np.ones((2, 96), bool)
    m[:, 32:64] = False
    m[1, :] = False                                   # batch 1: no visible key at all
    o, ms, ls = naive_attention(q, k, v, causal=False, kpad_mask=m, return_stats=True)
    o_t, ms_t, ls_t = tiled_flash_fwd(q, k, v, causal=False, kpad_mask=m, gsz=32)
    assert np.isfinite(o[0]).all() and np.isnan(o[1]).all()
    assert np.isfinite(o_t[0]).all() and np.isnan(o_t[1]).all()
    np.testing.assert_allclose(o_t[0], o[0], rtol=1e-10)
    assert (ls_t[1] == 0).all() and np.isneginf(ms_t[1]).all()


def test_check_messages_match_reference():
    """src/attention.jl:141-144"""
    z = lambda *s: np.zeros(s)
    with pytest.raises(ValueError, match="Embedding dim of Q `16` must be the same as of K `32`"):
        naive_attention(z(1, 1, 4, 16), z(1, 1, 4, 32), z(1, 1, 4, 32), causal=False)
    with pytest.raises(ValueError, match="must be the same"):
        naive_attention(z(1, 1, 4, 16), z(1, 1, 4, 16), z(1, 1, 5, 16), causal=False)
    with pytest.raises(ValueError, match="Only power-of-2 embedding dims"):
        naive_attention(z(1, 1, 4, 24), z(1, 1, 4, 24), z(1, 1, 4, 24), causal=False)
    with pytest.raises(ValueError, match="Number of query heads `3` must be divisible by number of KV heads `2`"):
        naive_attention(z(1, 3, 4, 16), z(1, 2, 4, 16), z(1, 2, 4, 16), causal=False)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[3:-4] for p in GOLDEN])
def test_golden_fixtures_reproduce(path):
    """The committed fixtures are exactly what the oracle computes today (regression pin)."""
    d = np.load(path)
    causal = bool(d["causal"])
    pair = d["pair"].astype(np.float64) if d["pair"].size else None
    mask = d["mask"] if d["mask"].size else None
    o, ms, ls = naive_attention(d["q"], d["k"], d["v"], pair, causal=causal, kpad_mask=mask, return_stats=True)
    dq, dk, dv, dp = naive_attention_grads(d["q"], d["k"], d["v"], d["do"], pair, causal=causal, kpad_mask=mask)
    for name, got in (("o", o), ("ms", ms), ("ls", ls), ("dq", dq), ("dk", dk), ("dv", dv)):
        np.testing.assert_allclose(got.astype(np.float32), d[name], rtol=2e-6, atol=1e-7, err_msg=name)
    if pair is not None:
        np.testing.assert_allclose(dp.astype(np.float32), d["dpair"], rtol=2e-6, atol=1e-7)
    assert len(GOLDEN) >= 8


def test_cpu_baseline_port_matches_oracle():
    """benchmarks/main.jl:26-43 restated in fp32 BLAS == fp64 oracle to fp32 accuracy."""
    q, k, v, do, _, _ = rand_case(11, 2, 2, 2, 96, 96, 64)
    for causal in (False, True):
        o = naive_attention(q, k, v, causal=causal)
        dq, dk, dv, _ = naive_attention_grads(q, k, v, do, causal=causal)
        o32 = naive_attention_f32(q, k, v, causal=causal)
        o32b, dq32, dk32, dv32 = naive_attention_f32_fwd_bwd(q, k, v, do, causal=causal)
        for a, b in ((o32, o), (o32b, o), (dq32, dq), (dk32, dk), (dv32, dv)):
            assert a.dtype == np.float32
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * np.abs(b).max())


def test_work_model_matches_survey_numbers():
    """SURVEY.md section 8(d) / BASELINE.md section 3."""
    assert attention_flops(64, 4096, 4096, 4, 4, causal=False) == 68_719_476_736
    assert attention_flops(64, 4096, 4096, 4, 4, causal=False, mode="bwd") == 171_798_691_840
    assert attention_flops(64, 4096, 4096, 4, 4, causal=False, mode="fwd+bwd") == 240_518_168_576
    assert attention_flops(128, 8192, 8192, 32, 8, causal=True) == 4_398_583_382_016
    lens = np.random.default_rng(0).integers(1024, 4097, size=16)
    assert list(lens[:4]) == [3637, 2981, 2594, 1853] and lens.sum() == 41_175
    assert attention_flops(128, 4096, 4096, 32, 16, causal=False, kv_lens=lens) == 2_763_207_475_200
    assert attention_bytes(64, 4096, 4096, 4, 4, 4, 2) == 33_816_576
    assert attention_bytes(64, 4096, 4096, 4, 4, 4, 4) == 67_633_152
    assert attention_bytes(128, 8192, 8192, 32, 32, 8, 2) == 2_155_872_256
    assert attention_bytes(128, 8192, 8192, 32, 32, 8, 2, mode="bwd") == 4_303_355_904


@pytest.mark.parametrize("causal", [False, True])
def test_slice_oracle_equals_full_oracle(causal):
    """naive_attention_slice (query-row chunks, used at the full BASELINE shapes) is the same formula as
    naive_attention / naive_attention_grads restricted to one (batch, kv-head) slice."""
    from oracle.naive_attention import naive_attention_slice
    rng = np.random.default_rng(5)
    B, QH, KH, L, E = 2, 4, 2, 300, 32
    q, do = rng.standard_normal((B, QH, L, E)), rng.standard_normal((B, QH, L, E))
    k, v = rng.standard_normal((B, KH, L, E)), rng.standard_normal((B, KH, L, E))
    m = np.ones((B, L), bool)
    m[1, -40:] = False
    o, ms, ls = naive_attention(q, k, v, causal=causal, kpad_mask=m, return_stats=True)
    dq, dk, dv, _ = naive_attention_grads(q, k, v, do, causal=causal, kpad_mask=m)
    r = naive_attention_slice(q[1, 2:4], k[1, 1], v[1, 1], do[1, 2:4], causal=causal, kpad_mask=m[1], chunk=64)
    for got, ref in ((r["o"], o[1, 2:4]), (r["ms"], ms[1, 2:4]), (r["ls"], ls[1, 2:4]), (r["dq"], dq[1, 2:4]),
                     (r["dk"], dk[1, 1]), (r["dv"], dv[1, 1])):
        np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-11)
