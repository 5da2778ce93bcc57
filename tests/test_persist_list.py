"""CPU: the static block list of the persistent kernels (csrc/fa_fwd_w64.hpp "persistent form", csrc/fa_bwd_w64.hpp, launch rule in
csrc/fa_fwd_inst.hpp launch_fwd_w64 / csrc/fa_bwd_inst.hpp launch_bwd_w64), restated here to pin its two properties: every block of
the problem is visited exactly once, and under a causal mask every workgroup gets the same work.  (That the KERNELS walk this list is
what the GPU tests check: bitwise equality with the one-block-per-workgroup launch, tests/test_fwd_w64_gpu.py / test_bwd_w64_gpu.py.)"""
import numpy as np
import pytest


def block_of(wg, step, n_blk, B, H, hx, first_heaviest):
    """(batch*H + head, block) of workgroup `wg` (0..255) at `step` -- the decode at the top of the kernels' block loop."""
    x, c = wg & 7, wg >> 3
    pos = 32 * step + (31 - c if step & 1 else c)
    col, k = divmod(pos, n_blk)
    bh = (col // hx) * H + x * hx + col % hx if hx > 0 else x * ((B * H) >> 3) + col
    blk = k if first_heaviest else n_blk - 1 - k      # dK/dV under the causal mask: the first key block sees every query; forward / dQ: the last query block sees every key
    return bh, blk


def steps_of(n_blk, B, H):
    """the launcher's rule: None = the list does not divide (one block per workgroup instead)"""
    bh = B * H
    per_xcd = (bh // 8) * n_blk
    if bh % 8 or n_blk & (n_blk - 1) or per_xcd % 32 or per_xcd // 32 < 2:
        return None
    return per_xcd // 32


@pytest.mark.parametrize("name,L,rows,B,H", [("C3 forward / dQ", 8192, 256, 8, 32), ("C5 shard", 16384, 256, 8, 32), ("C4 forward", 4096, 256, 16, 32),
                                             ("C4 dK/dV (kv heads)", 4096, 256, 16, 8), ("GQA 16/8, 4 batches", 4096, 256, 4, 8),
                                             ("E = 256 dQ (128-row blocks)", 2048, 128, 4, 16), ("short", 2048, 256, 4, 16)])
@pytest.mark.parametrize("batch_major", [False, True])
@pytest.mark.parametrize("first_heaviest", [False, True])
def test_every_block_once_and_equal_causal_work(name, L, rows, B, H, batch_major, first_heaviest):
    n_blk = L // rows
    S = steps_of(n_blk, B, H)
    assert S is not None, name
    hx = H // 8 if (batch_major and H % 8 == 0) else 0
    seen = np.zeros((B * H, n_blk), dtype=np.int32)
    work = np.zeros(256, dtype=np.int64)
    per_batch = np.zeros((256, B), dtype=np.int64)
    for wg in range(256):
        for s in range(S):
            bh, blk = block_of(wg, s, n_blk, B, H, hx, first_heaviest)
            seen[bh, blk] += 1
            work[wg] += (n_blk - blk) if first_heaviest else (blk + 1)        # causal: tiles the block walks
            per_batch[wg, bh // H] += 1
    assert (seen == 1).all(), name                                             # a bijection onto the blocks of the problem
    if S % 2 == 0:
        assert work.min() == work.max(), (name, work.min(), work.max())        # causal: the same work for every workgroup
    else:
        assert work.max() - work.min() <= n_blk
    if hx > 0 and n_blk <= 32:
        # batch-major columns: every workgroup sees every batch equally often (per-batch key lengths then weigh on all alike)
        assert (per_batch == per_batch[0]).all() or per_batch.std(axis=0).max() <= 1.0


def test_rule_rejects_what_does_not_divide():
    assert steps_of(16, 4, 4) is None          # C2: 256 blocks = one round, nothing to walk
    assert steps_of(24, 8, 32) is None         # q-blocks not a power of two
    assert steps_of(32, 3, 4) is None          # (batch x head) columns not in eighths
