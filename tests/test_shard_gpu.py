"""Multi-GPU partition driven by the HIP operator on ONE GPU: every rank's share of `shard.flash_attention_sharded`
(forward, and backward through autograd over the rectangle views) and of `shard.flash_attention_sharded_fwd_bwd` is
computed in a loop over ranks and must reproduce the unsharded launch BITWISE -- a (batch, kv-head) unit's arithmetic
does not depend on which other units share its launch (src/attention.jl:27-28,33; src/attention_bwd.jl:28-29,34), and
gradients shard like their inputs with no cross-rank reduction.  world in {2, 3, 8}; GQA; unit counts that do not divide
(C2 on 8 GPUs = 2 units per GPU, B < G splits over kv heads)."""
import pytest
import torch

from util import make_inputs

pytestmark = pytest.mark.gpu


def _eq(a, b):
    return torch.equal(torch.nan_to_num(a.float()), torch.nan_to_num(b.float()))


CASES = [
    # B, QH, KH, QL, KL, E, dt, causal, pad
    (4, 4, 4, 512, 512, 64, "bf16", False, None),        # C2's unit structure (16 units): 8 / 5-6 / 2 per rank
    (2, 8, 2, 384, 320, 64, "f16", True, "ref"),         # GQA 8/2: 4 units, world 3 and 8 leave ranks uneven / empty
    (3, 6, 3, 300, 300, 128, "bf16", True, None),        # 9 units, E = 128, ragged tiles
    (1, 6, 3, 257, 257, 32, "f32", False, "lens"),       # B < world: the split runs over kv heads
]


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"B{c[0]}H{c[1]}-{c[2]}L{c[3]}E{c[5]}{c[6]}{'c' if c[7] else ''}")
def test_sharded_hip_forward_backward_is_bitwise_the_unsharded_launch(pkg, dev, case, world):
    B, QH, KH, QL, KL, E, dt, causal, pad = case
    d = make_inputs(61, B, QH, KH, QL, KL, E, dt, dev, pad=pad)
    rep = QH // KH
    # unsharded reference launch
    q, k, v = (d[n].clone().requires_grad_(True) for n in ("q", "k", "v"))
    o_full = pkg.flash_attention(q, k, v, causal=causal, kpad_mask=d["mask"])
    o_full.backward(d["do"])
    ref = dict(o=o_full.detach(), dq=q.grad, dk=k.grad, dv=v.grad)

    # (1) autograd through the rectangle views, rank by rank
    q2, k2, v2 = (d[n].clone().requires_grad_(True) for n in ("q", "k", "v"))
    do_units = d["do"].reshape(B * KH, rep, QL, E)
    seen = 0
    for rank in range(world):
        lo, hi = pkg.shard.unit_range(B * KH, world, rank)
        local = pkg.shard.flash_attention_sharded(q2, k2, v2, causal=causal, kpad_mask=d["mask"], world=world, rank=rank)
        assert tuple(local.shape) == (hi - lo, rep, QL, E)
        assert _eq(local, ref["o"].reshape(B * KH, rep, QL, E)[lo:hi]), f"forward, rank {rank}"
        if hi > lo:
            local.backward(do_units[lo:hi])
        seen += hi - lo
    assert seen == B * KH
    for name, t in (("dq", q2.grad), ("dk", k2.grad), ("dv", v2.grad)):
        assert _eq(t, ref[name]), f"{name} through autograd over the shard views"

    # (2) the explicit forward+backward driver: results are the rectangles, sharded like the inputs
    for rank in range(world):
        for rect, o, dq, dk, dv, _ in pkg.shard.flash_attention_sharded_fwd_bwd(
                d["q"], d["k"], d["v"], d["do"], causal=causal, kpad_mask=d["mask"], world=world, rank=rank):
            hs = slice(rect.kh0 * rep, rect.kh1 * rep)
            assert _eq(o, ref["o"][rect.b0:rect.b1, hs])
            assert _eq(dq, ref["dq"][rect.b0:rect.b1, hs])
            assert _eq(dk, ref["dk"][rect.b0:rect.b1, rect.kh0:rect.kh1])
            assert _eq(dv, ref["dv"][rect.b0:rect.b1, rect.kh0:rect.kh1])
