"""CPU tests of the host-side mirror (no GPU): GPU-only contract, argument errors, sharding."""
import numpy as np
import pytest
import torch


def test_product_path_refuses_cpu_tensors(pkg):
    """cpu=false in the reference (src/attention.jl:1); no CPU / PyTorch fallback here either."""
    q = torch.zeros(1, 1, 8, 16)
    with pytest.raises(pkg.NNopError, match="GPU-only"):
        pkg.flash_attention(q, q, q, causal=False)
    with pytest.raises(pkg.NNopError, match="GPU-only"):
        pkg._flash_attention(q, q, q, causal=False)


def test_causal_is_a_required_keyword(pkg):
    q = torch.zeros(1, 1, 8, 16)
    with pytest.raises(TypeError):
        pkg.flash_attention(q, q, q)


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libnnop_hip.so")
    with pytest.raises(ImportError, match="no CPU or PyTorch fallback"):
        pkg._lib.load()


def test_product_package_never_imports_the_oracle():
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "nnop.jl_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".jl")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"


@pytest.mark.parametrize("B,KH,world", [(4, 4, 8), (4, 4, 2), (64, 32, 8), (3, 2, 4), (5, 3, 4), (1, 1, 2), (2, 8, 3)])
def test_rectangles_partition_units_exactly_once(pkg, B, KH, world):
    seen = np.zeros((B, KH), int)
    sizes = []
    for rank in range(world):
        rects = pkg.shard.rectangles(B, KH, world, rank)
        assert len(rects) <= 3
        n = 0
        for r in rects:
            seen[r.b0:r.b1, r.kh0:r.kh1] += 1
            n += r.units
        lo, hi = pkg.shard.unit_range(B * KH, world, rank)
        assert n == hi - lo
        sizes.append(n)
    assert (seen == 1).all()
    assert max(sizes) - min(sizes) <= 1           # balanced


def test_headline_config_sharding(pkg):
    """C2 on 8 GPUs: B=4, KH=4 -> 2 (batch, head) slices per GPU; C5: 8 whole batches per GPU."""
    assert [r.units for r in pkg.shard.rectangles(4, 4, 8, 3)] == [2]
    r = pkg.shard.rectangles(64, 32, 8, 5)
    assert len(r) == 1 and (r[0].b0, r[0].b1, r[0].kh0, r[0].kh1) == (40, 48, 0, 32)


def test_shard_views_are_views_and_gqa_heads_stay_together(pkg):
    q = torch.arange(2 * 8 * 3 * 4, dtype=torch.float32).reshape(2, 8, 3, 4)
    k = torch.arange(2 * 2 * 5 * 4, dtype=torch.float32).reshape(2, 2, 5, 4)
    rect = pkg.shard.Rect(1, 2, 1, 2)
    qs, ks, vs, ps, ms = pkg.shard.shard_views(rect, q, k, k)
    assert qs.shape == (1, 4, 3, 4) and ks.shape == (1, 1, 5, 4)
    assert qs.data_ptr() == q[1, 4].data_ptr() and ks.data_ptr() == k[1, 1].data_ptr()
    assert qs.is_contiguous() and ks.is_contiguous()


def test_graft_entry_build_runs():
    """__graft_entry__.build() (what the driver calls every round) compiles, loads and checks the ABI version."""
    import __graft_entry__ as g
    g.build()


def test_workmodel_matches_the_oracles_work_model(pkg):
    """bench.py and the perf tools take FLOPs / bytes from the product's workmodel; the oracles carry the same
    formulas for the tests -- they must agree."""
    from oracle.naive_attention import attention_bytes, attention_flops
    from oracle.naive_norms import norm_bytes
    from oracle.naive_rope import rope_bytes
    from oracle.naive_softmax import softmax_bytes
    wm = pkg.workmodel
    for causal in (False, True):
        for mode in ("fwd", "bwd", "fwd+bwd"):
            assert wm.attention_flops(64, 4096, 4096, 4, 4, causal=causal, mode=mode) == \
                attention_flops(64, 4096, 4096, 4, 4, causal=causal, mode=mode)
            assert wm.attention_bytes(128, 300, 500, 8, 2, 3, 2, mode=mode) == attention_bytes(128, 300, 500, 8, 2, 3, 2, mode=mode)
    assert wm.attention_flops(128, 4096, 4096, 32, 3, causal=False, kv_lens=[100, 4096, 7]) == \
        attention_flops(128, 4096, 4096, 32, 3, causal=False, kv_lens=[100, 4096, 7])
    assert wm.attention_flops(64, 4096, 4096, 4, 4, causal=False) == 4 * 64 * 4096 * 4096 * 16
    assert wm.rope_bytes(128, 4096, 32, 8, 2, 2) == rope_bytes(128, 4096, 32, 8, 2, 2)
    for bwd in (False, True):
        assert wm.softmax_bytes(1024, 7, 4, bwd=bwd) == softmax_bytes(1024, 7, 4, bwd=bwd)
        assert wm.norm_bytes(1024, 7, 2, bwd=bwd) == norm_bytes(1024, 7, 2, bwd=bwd)


def test_rectangles_cover_every_unit_once(pkg):
    for B, KH in ((4, 4), (64, 32), (1, 3), (5, 7)):
        for world in (1, 2, 3, 8):
            got = []
            for rank in range(world):
                for r in pkg.shard.rectangles(B, KH, world, rank):
                    got += [b * KH + kh for b in range(r.b0, r.b1) for kh in range(r.kh0, r.kh1)]
            assert got == list(range(B * KH))
