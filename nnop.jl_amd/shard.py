"""Multi-GPU sharding of the attention hot path: one process per GPU, no data-path collective.

Every (batch, kv-head) slice -- a KV head together with its QH/KH query heads -- is independent in
the forward and in the backward (``gidx[2], gidx[3]`` only index, never mix:
src/attention.jl:27-28,33,126-129; src/attention_bwd.jl:28-29,34), so the H x B axis shards
embarrassingly.  The reference itself has no multi-GPU code.

Units are numbered u = b * KH + kh (batch slowest, as in memory); rank g owns the contiguous
range [g*U/G, (g+1)*U/G).  Because the layout is [B][H][L][E], a unit range is at most three
dense rectangles (tail of the first batch, whole batches, head of the last batch), each of which
is a plain ``[B', H', L, E]`` problem for the kernels -- pointer offsets only, no copies.
Gradients are sharded exactly like their inputs (dK/dV of a kv head need all of its query heads,
which is why a kv head's query heads never split across ranks).

The only collective is optional result replication (``gather=True``): RCCL all-gather over xGMI
(``torch.distributed`` backend "nccl" IS RCCL on ROCm; "gloo" on CPU in the tests).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional

import torch

__all__ = ["Rect", "unit_range", "rectangles", "shard_views", "flash_attention_sharded",
           "flash_attention_sharded_fwd_bwd", "all_gather_units"]


@dataclass(frozen=True)
class Rect:
    """Batches [b0, b1) x kv-heads [kh0, kh1)."""
    b0: int
    b1: int
    kh0: int
    kh1: int

    @property
    def units(self) -> int:
        return (self.b1 - self.b0) * (self.kh1 - self.kh0)


def unit_range(n_units: int, world: int, rank: int):
    """Contiguous, balanced split of `n_units` over `world` ranks: [lo, hi)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (n_units * rank) // world, (n_units * (rank + 1)) // world


def rectangles(B: int, KH: int, world: int, rank: int) -> List[Rect]:
    """Dense rectangles covering rank's unit range (at most three)."""
    lo, hi = unit_range(B * KH, world, rank)
    out: List[Rect] = []
    u = lo
    while u < hi:
        b, kh = divmod(u, KH)
        if kh == 0 and hi - u >= KH:                 # whole batches
            nb = (hi - u) // KH
            out.append(Rect(b, b + nb, 0, KH))
            u += nb * KH
        else:                                        # part of one batch
            kh1 = min(KH, kh + (hi - u))
            out.append(Rect(b, b + 1, kh, kh1))
            u += kh1 - kh
    return out


def shard_views(rect: Rect, q, k, v, pair=None, kpad_mask=None):
    """Views (no copies, except a head-sliced `pair`) of the tensors restricted to `rect`."""
    rep = q.shape[1] // k.shape[1]
    qs = q[rect.b0:rect.b1, rect.kh0 * rep:rect.kh1 * rep]
    ks = k[rect.b0:rect.b1, rect.kh0:rect.kh1]
    vs = v[rect.b0:rect.b1, rect.kh0:rect.kh1]
    ps = None
    if pair is not None:
        ps = pair[rect.b0:rect.b1, :, :, rect.kh0 * rep:rect.kh1 * rep]
    ms = kpad_mask[rect.b0:rect.b1] if kpad_mask is not None else None
    return qs, ks, vs, ps, ms


def all_gather_units(local: torch.Tensor, n_units: int, group=None) -> torch.Tensor:
    """Replicate a unit-sharded tensor ``[units_local, ...]`` on every rank -> ``[n_units, ...]``.

    One all-gather; shards may differ in size by one unit (padded to the largest)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = [unit_range(n_units, world, r)[1] - unit_range(n_units, world, r)[0] for r in range(world)]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
    if len(set(sizes)) == 1:
        out = torch.empty((n_units,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
        return out
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous(), group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def forward_with_overlapped_gather(step_chunk, o_chunks, full_chunks, group=None):
    """The optional replication of O overlapped with compute (last hard part): the per-GPU batch is cut into
    chunks along the batch axis; chunk i's all-gather is issued asynchronously right behind its forward launch -- RCCL runs it on
    its own stream, ordered behind the launch by an event -- while chunk i+1 computes on the launch stream.

    step_chunk(i): launches the forward of chunk i on the current stream (writes o_chunks[i]);
    full_chunks[i]: the destination, rank r's chunk at rows [r * n, (r + 1) * n) for n = o_chunks[i].shape[0] (the concatenated form
    both RCCL and gloo accept).  Returns after every gather has completed on the current stream."""
    import torch.distributed as dist
    works = []
    for i in range(len(o_chunks)):
        step_chunk(i)
        works.append(dist.all_gather_into_tensor(full_chunks[i], o_chunks[i], group=group, async_op=True))
    for w in works:
        w.wait()


def flash_attention_sharded(q, k, v, pair=None, *, causal: bool, kpad_mask=None,
                            world: Optional[int] = None, rank: Optional[int] = None,
                            gather: bool = False, group=None,
                            attn_fn: Optional[Callable] = None):
    """Run this rank's share of ``flash_attention(q,k,v,pair; causal,kpad_mask)``.

    Inputs are the FULL (replicated) tensors; the rank computes only its (batch, kv-head)
    rectangles.  Returns the local output as ``[units_local * rep, QL, E]``-shaped rows in unit
    order (``gather=False``) or the full ``[B, QH, QL, E]`` output replicated by one all-gather
    (``gather=True``).  ``attn_fn`` defaults to the HIP operator and exists so that the CPU
    (gloo) tests can exercise the partition + collective plumbing with a stand-in.
    """
    if attn_fn is None:
        from .attention import flash_attention as attn_fn      # the HIP path; no fallback
    if world is None or rank is None:
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    B, QH, QL, E = q.shape
    KH = k.shape[1]
    rep = QH // KH
    outs = []
    for rect in rectangles(B, KH, world, rank):
        qs, ks, vs, ps, ms = shard_views(rect, q, k, v, pair, kpad_mask)
        o = attn_fn(qs, ks, vs, ps, causal=causal, kpad_mask=ms)
        outs.append(o.reshape(rect.units, rep, QL, E))
    if outs:
        local = torch.cat(outs, dim=0) if len(outs) > 1 else outs[0]
    else:
        local = q.new_empty((0, rep, QL, E))
    if not gather:
        return local
    full = all_gather_units(local, B * KH, group=group)
    return full.reshape(B, QH, QL, E)


def flash_attention_sharded_fwd_bwd(q, k, v, dO, pair=None, *, causal: bool, kpad_mask=None,
                                    world: int, rank: int):
    """This rank's share of one forward + backward, without autograd: for each of the rank's rectangles
    ``_flash_attention`` then ``grad_flash_attention`` on the rectangle's views (pointer offsets only).

    Gradients are sharded exactly like their inputs -- dq like q, dk / dv like k / v -- and need no
    cross-rank reduction: a kv head's query heads never split across ranks (module docstring).
    Returns a list of ``(rect, o, dq, dk, dv, dpair)`` in unit order; the tensors are the rectangle-shaped
    results ([B', H', L, E]).  The HIP operator is the only implementation (no fallback).
    """
    from .attention import _flash_attention, grad_flash_attention
    rep = q.shape[1] // k.shape[1]
    out = []
    for rect in rectangles(q.shape[0], k.shape[1], world, rank):
        qs, ks, vs, ps, ms_ = shard_views(rect, q, k, v, pair, kpad_mask)
        dos = dO[rect.b0:rect.b1, rect.kh0 * rep:rect.kh1 * rep]
        o, m, l = _flash_attention(qs, ks, vs, ps, causal=causal, kpad_mask=ms_)
        dq, dk, dv, dp = grad_flash_attention(dos, o, m, l, qs, ks, vs, ps, causal=causal, kpad_mask=ms_)
        out.append((rect, o, dq, dk, dv, dp))
    return out
