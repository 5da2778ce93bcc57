"""GPU tests of the operator surface: error behaviour with the reference's messages
(src/attention.jl:141-144), residual contract, asynchronous launch on the current stream,
nnop_shared_memory, re-entrancy from two host threads on two streams."""
import threading

import numpy as np
import pytest
import torch

from util import make_inputs, oracle_fwd, assert_close

pytestmark = pytest.mark.gpu


def z(dev, *s, dt=torch.bfloat16):
    return torch.zeros(*s, dtype=dt, device=dev)


def test_reference_error_messages(pkg, dev):
    with pytest.raises(pkg.NNopError, match=r"Embedding dim of Q `16` must be the same as of K `32`\."):
        pkg.flash_attention(z(dev, 1, 1, 8, 16), z(dev, 1, 1, 8, 32), z(dev, 1, 1, 8, 32), causal=False)
    with pytest.raises(pkg.NNopError, match=r"Shapes of K `\(16, 8, 1, 1\)` and V `\(16, 9, 1, 1\)` must be the same\."):
        pkg.flash_attention(z(dev, 1, 1, 8, 16), z(dev, 1, 1, 8, 16), z(dev, 1, 1, 9, 16), causal=False)
    with pytest.raises(pkg.NNopError, match="Only power-of-2 embedding dims are supported."):
        pkg.flash_attention(z(dev, 1, 1, 8, 24), z(dev, 1, 1, 8, 24), z(dev, 1, 1, 8, 24), causal=False)
    with pytest.raises(pkg.NNopError, match="Number of query heads `3` must be divisible by number of KV heads `2`."):
        pkg.flash_attention(z(dev, 1, 3, 8, 16), z(dev, 1, 2, 8, 16), z(dev, 1, 2, 8, 16), causal=False)
    with pytest.raises(pkg.NNopError, match="Failed to find groupsize"):
        pkg.flash_attention(z(dev, 1, 1, 8, 1024), z(dev, 1, 1, 8, 1024), z(dev, 1, 1, 8, 1024), causal=False)   # powers of two up to 512 run
    with pytest.raises(TypeError):          # mixed element types: MethodError in the reference
        pkg.flash_attention(z(dev, 1, 1, 8, 16), z(dev, 1, 1, 8, 16, dt=torch.float16), z(dev, 1, 1, 8, 16), causal=False)
    with pytest.raises(TypeError):
        pkg.flash_attention(z(dev, 1, 1, 8, 16, dt=torch.float64), z(dev, 1, 1, 8, 16, dt=torch.float64),
                            z(dev, 1, 1, 8, 16, dt=torch.float64), causal=False)


def test_shared_memory_hook(pkg, dev):
    """ext/NNopAMDGPUExt.jl:6-9: sharedMemPerBlock of the device."""
    n = pkg.shared_memory(0)
    assert n == torch.cuda.get_device_properties(0).shared_memory_per_block
    assert n >= 64 * 1024


def test_residuals_dtype_shape_and_meaning(pkg, dev):
    """ms, ls are (QL,QH,B) arrays of T (src/attention.jl:167-168): ms = row max, ls = sum exp(s - ms)."""
    for dt in ("f32", "bf16"):
        d = make_inputs(21, 2, 2, 2, 130, 130, 32, dt, dev, need_do=False)
        o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=True)
        assert ms.dtype == d["q"].dtype and ls.dtype == d["q"].dtype
        assert ms.shape == (2, 2, 130) and ls.shape == (2, 2, 130) and o.shape == d["q"].shape
        _, ms_ref, ls_ref = oracle_fwd(d, True)
        assert_close("ms", ms, ms_ref, dt)
        assert_close("ls", ls, ls_ref, dt, 2.0)
        assert (ls.float() >= 1.0 - 1e-2).all()      # the max element contributes exp(0) = 1


def test_dead_rows_are_nan_like_the_naive_formula(pkg, dev):
    """A row with no visible key: 0/0 = NaN in the naive formula and in the reference; a fully masked
    K TILE alone must stay finite."""
    d = make_inputs(22, 2, 2, 2, 64, 192, 64, "f32", dev, need_do=False)
    m = torch.ones(2, 192, dtype=torch.bool, device=dev)
    m[:, 64:128] = False
    m[1, :] = False
    d["mask"] = m
    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=False, kpad_mask=m)
    torch.cuda.synchronize()
    o_ref, _, _ = oracle_fwd(d, False)
    assert torch.isfinite(o[0]).all() and torch.isnan(o[1]).all()
    assert_close("o", o, o_ref, "f32")
    assert (ls[1] == 0).all() and torch.isneginf(ms[1]).all()


def test_async_on_current_stream_and_thread_reentrancy(pkg, dev):
    """The op launches on the caller's stream and never synchronises (src/attention.jl:170-176);
    concurrent calls from two host threads on two streams give the single-threaded result."""
    d1 = make_inputs(23, 2, 4, 4, 512, 512, 64, "bf16", dev)
    d2 = make_inputs(24, 2, 4, 2, 384, 384, 128, "f16", dev)
    ref1 = pkg._flash_attention(d1["q"], d1["k"], d1["v"], causal=True)[0].clone()
    ref2 = pkg._flash_attention(d2["q"], d2["k"], d2["v"], causal=False)[0].clone()
    torch.cuda.synchronize()
    out, errs = {}, []

    def work(name, d, causal, n):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                for _ in range(n):
                    o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal)
                    g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal)
                s.synchronize()
            out[name] = (o, g)
        except Exception as e:                       # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=("a", d1, True, 8)), threading.Thread(target=work, args=("b", d2, False, 8))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert torch.equal(out["a"][0], ref1) and torch.equal(out["b"][0], ref2)     # deterministic, bitwise
    g1 = pkg.grad_flash_attention(d1["do"], *pkg._flash_attention(d1["q"], d1["k"], d1["v"], causal=True),
                                  d1["q"], d1["k"], d1["v"], causal=True)
    for a, b in zip(out["a"][1][:3], g1[:3]):
        assert torch.equal(a, b)                    # no atomics anywhere: gradients are bitwise reproducible


def test_noncontiguous_inputs_are_accepted(pkg, dev):
    d = make_inputs(25, 2, 2, 2, 100, 100, 32, "f32", dev, need_do=False)
    qt = d["q"].transpose(1, 2).contiguous().transpose(1, 2)      # same values, permuted strides
    assert not qt.is_contiguous()
    a = pkg.flash_attention(qt, d["k"], d["v"], causal=False)
    b = pkg.flash_attention(d["q"], d["k"], d["v"], causal=False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dt", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("E,causal,pad", [(64, False, None), (64, True, None), (64, False, "ref"), (128, True, None),
                                          (32, True, "lens"), (16, False, None)])
def test_bitwise_reproducible_across_launches_and_workgroup_shapes(pkg, dev, dt, E, causal, pad, tune):
    """No atomics and no cross-workgroup reduction anywhere: repeated launches are bitwise identical,
    also right after the caches were flushed, and the 4-wave and 8-wave forward variants (same per-row
    arithmetic, different workgroup shape) agree bitwise.  Guards against races / hazards that a
    tolerance check can miss."""
    d = make_inputs(31, 2, 4, 2, 517, 517, E, dt, dev, pad=pad)
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
    outs = []
    tune(fwd_split=0, fwd_w64=0)   # compare the 8-wave and 4-wave forms of the same per-row arithmetic
    for nw in (8, 4, 4, 8):
        tune(fwd_nw=nw)
        flush.fill_(1)
        o, ms, ls = pkg._flash_attention(d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
        g = pkg.grad_flash_attention(d["do"], o, ms, ls, d["q"], d["k"], d["v"], causal=causal, kpad_mask=d["mask"])
        torch.cuda.synchronize()
        outs.append((o, ms, ls) + tuple(g[:3]))
    for other in outs[1:]:
        for a, b, name in zip(outs[0], other, ("o", "ms", "ls", "dq", "dk", "dv")):
            assert torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float())), name


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("E,QL,KL", [(64, 1024, 1024), (64, 300, 192), (32, 512, 576), (16, 257, 128), (64, 4096, 4096)])
def test_split_kv_form_matches_plain_form_and_is_reproducible(pkg, dev, dt, E, QL, KL, tune):
    """The 16-wave split-KV forward (default in plain mode) sums the keys in a different order than the 8-wave
    form, so the two agree to rounding, not bitwise; each is bitwise reproducible; both match the oracle."""
    d = make_inputs(33, 2, 2, 2, QL, KL, E, dt, dev, need_do=False)
    tune(fwd_split=1, fwd_w64=0)
    a = pkg._flash_attention(d["q"], d["k"], d["v"], causal=False)
    b = pkg._flash_attention(d["q"], d["k"], d["v"], causal=False)
    tune(fwd_split=0)
    c = pkg._flash_attention(d["q"], d["k"], d["v"], causal=False)
    torch.cuda.synchronize()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    if QL * KL <= 1024 * 1024:
        o_ref, ms_ref, ls_ref = oracle_fwd(d, False)
        for res in (a, c):
            assert_close("o", res[0], o_ref, dt)
            assert_close("ms", res[1], ms_ref, dt)
            lse = res[1].double().cpu().numpy() + np.log(res[2].double().cpu().numpy())
            assert_close("lse", lse, ms_ref + np.log(ls_ref), dt)
    else:
        rel = float((a[0].float() - c[0].float()).abs().max() / c[0].float().abs().max())
        assert rel < 1e-2


def test_hip_graph_capture_and_replay(pkg, dev):
    """The C ABI never allocates, synchronises or touches the default stream, so a forward + backward (and the row
    operators) capture into a HIP graph; replaying the graph on new data in the static buffers reproduces the eager
    results bitwise.  This is the launch path for launch-bound shapes (a graph replay costs one submission)."""
    dt = torch.bfloat16
    B, QH, KH, L, E = 2, 4, 2, 384, 64
    g = torch.Generator(device=dev).manual_seed(7)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
    q, k, v, do = mk(B, QH, L, E), mk(B, KH, L, E), mk(B, KH, L, E), mk(B, QH, L, E)
    o = torch.empty_like(q); ms = torch.empty(B, QH, L, dtype=dt, device=dev); ls = torch.empty_like(ms)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=True), dtype=torch.uint8, device=dev)
    x = mk(64, 1024); y = torch.empty_like(x); w = torch.ones(1024, device=dev)

    def step():
        pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=True)
        pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=True)
        pkg.online_softmax_into(y, x)

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        step()                                            # warm-up outside capture (sets kernel attributes)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    results = []
    for seed in (1, 2):
        g2 = torch.Generator(device=dev).manual_seed(seed)
        for t in (q, k, v, do, x):
            t.copy_(torch.randn(t.shape, device=dev, generator=g2).to(dt))
        graph.replay()
        torch.cuda.synchronize()
        replayed = [t.clone() for t in (o, ms, ls, dq, dk, dv, y)]
        step()
        torch.cuda.synchronize()
        for a, b_ in zip(replayed, (o, ms, ls, dq, dk, dv, y)):
            assert torch.equal(a, b_)
        results.append(replayed[0])
    assert not torch.equal(results[0], results[1])       # the replays really consumed the new data
