"""Generates the golden fixtures in this directory from the fp64 oracle.

    python tests/golden/make_golden.py

The reference (Julia, GPU-only kernels, no golden vectors of its own -- SURVEY.md section 8(c))
cannot produce vectors here, so these fixtures freeze the ORACLE: inputs are N(0,1) from
numpy.random.default_rng(seed), rounded to bf16-representable fp32 values (so the same fixture
is exact input for the f32, f16 and bf16 paths), expected outputs are the fp64 oracle's, stored
as fp32.  A fixture is data only: inputs and expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.naive_attention import naive_attention, naive_attention_grads  # noqa: E402
from oracle.naive_rope import llama_rotary_embedding, naive_llama_rope, pairwise_llama_rope  # noqa: E402
from oracle.naive_softmax import naive_softmax, naive_softmax_grad  # noqa: E402
from oracle.naive_norms import (naive_rms_norm, naive_rms_norm_grads, naive_layer_norm,  # noqa: E402
                                naive_layer_norm_grads)

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (seed, B, QH, KH, QL, KL, E, causal, pad, pair)
CASES = {
    "plain_e64":        (0, 2, 2, 2, 128, 128, 64, False, None, False),
    "ragged_e32":       (1, 2, 2, 2, 95, 130, 32, False, None, False),
    "causal_e16":       (2, 2, 2, 2, 97, 97, 16, True, None, False),
    "causal_pad_e64":   (3, 2, 2, 2, 100, 100, 64, True, "ref", False),
    "gqa_causal_e32":   (4, 2, 6, 2, 65, 65, 32, True, None, False),
    "pair_pad_e16":     (5, 2, 2, 2, 70, 81, 16, False, "ref", True),
    "varlen_e128":      (6, 3, 4, 1, 80, 144, 128, False, "lens", False),
    "deadtile_e64":     (7, 2, 2, 2, 64, 200, 64, False, "deadtile", False),
}


def bf16_round(x):
    """Round fp32 to the nearest bf16-representable fp32 (round-to-nearest-even)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def make_case(seed, B, QH, KH, QL, KL, E, causal, pad, pair):
    rng = np.random.default_rng(seed)
    g = lambda *s: bf16_round(rng.standard_normal(s).astype(np.float32))
    d = dict(q=g(B, QH, QL, E), k=g(B, KH, KL, E), v=g(B, KH, KL, E), do=g(B, QH, QL, E))
    d["pair"] = g(B, KL, QL, QH) if pair else np.zeros((0,), np.float32)
    mask = np.ones((B, KL), dtype=bool)
    if pad == "ref":            # test/attention_tests.jl:27-28
        mask[-1, -11:] = False
    elif pad == "lens":         # variable sequence lengths
        lens = rng.integers(KL // 4, KL + 1, size=B)
        mask = np.arange(KL)[None, :] < lens[:, None]
    elif pad == "deadtile":     # a whole 64-key tile masked (the reference NaNs here; the naive formula does not)
        mask[:, 64:128] = False
        mask[-1, 150:] = False
    d["mask"] = mask if pad is not None else np.zeros((0,), bool)
    p = d["pair"].astype(np.float64) if pair else None
    m = mask if pad is not None else None
    o, ms, ls = naive_attention(d["q"], d["k"], d["v"], p, causal=causal, kpad_mask=m, return_stats=True)
    dq, dk, dv, dp = naive_attention_grads(d["q"], d["k"], d["v"], d["do"], p, causal=causal, kpad_mask=m)
    d.update(o=o.astype(np.float32), ms=ms.astype(np.float32), ls=ls.astype(np.float32),
             dq=dq.astype(np.float32), dk=dk.astype(np.float32), dv=dv.astype(np.float32),
             dpair=(dp.astype(np.float32) if pair else np.zeros((0,), np.float32)),
             causal=np.array(causal))
    return d


# Llama RoPE.  name: (seed, B, QH, KH, L, D, position offset)
ROPE_CASES = {
    "ref_d16":   (20, 2, 3, 5, 257, 16, 0),        # the reference's test dim (test/rope_tests.jl:22)
    "gqa_d128":  (21, 2, 8, 2, 131, 128, 1000),
    "odd_d6":    (22, 1, 2, 1, 40, 6, 7),          # half dim not a multiple of 8: generic path
}


def make_rope_case(seed, B, QH, KH, L, D, off):
    rng = np.random.default_rng(seed)
    g = lambda *s: bf16_round(rng.standard_normal(s).astype(np.float32))
    pos = np.tile(np.arange(L, dtype=np.float32) + off, (B, 1))
    cos, sin = llama_rotary_embedding(D, pos)                       # fp32 tables, stored as the kernel receives them
    d = dict(q=g(B, QH, L, D), k=g(B, KH, L, D), dq_out=g(B, QH, L, D), dk_out=g(B, KH, L, D), cos=cos, sin=sin,
             position_ids=pos)
    qo, ko = naive_llama_rope(d["q"], d["k"], cos, sin)
    dq, dk = pairwise_llama_rope(d["dq_out"], d["dk_out"], cos, sin, sin_sign=-1.0)
    d.update(q_out=qo.astype(np.float32), k_out=ko.astype(np.float32), dq=dq.astype(np.float32), dk=dk.astype(np.float32))
    return d


# Row operators.  name: (seed, n, emb): one matrix serves softmax, RMSNorm and LayerNorm
ROW_CASES = {
    "wave_520":     (30, 23, 520),       # wave-per-row register shape, partial last chunk
    "odd_257":      (31, 17, 257),       # odd length: generic kernels (reference grid value)
    "block_4096":   (32, 9, 4096),       # workgroup-per-row shape
}


def make_row_case(seed, n, emb):
    rng = np.random.default_rng(seed)
    g = lambda *s: bf16_round(rng.standard_normal(s).astype(np.float32))
    d = dict(x=g(n, emb), dy=g(n, emb), w=g(emb), b=g(emb), offset=np.float32(0.5), eps=np.float32(1e-5))
    y = naive_softmax(d["x"])
    d["softmax_y"] = y.astype(np.float32)
    # the pullback is taken at the fp32-rounded y stored here (what a caller would hand back)
    d["softmax_dx"] = naive_softmax_grad(d["dy"], d["softmax_y"]).astype(np.float32)
    yr, rstd = naive_rms_norm(d["x"], d["w"], offset=0.5, eps=1e-5)
    dxr, dwr = naive_rms_norm_grads(d["dy"], d["x"], d["w"], offset=0.5, eps=1e-5)
    d.update(rms_y=yr.astype(np.float32), rms_rstd=rstd.astype(np.float32), rms_dx=dxr.astype(np.float32),
             rms_dw=dwr.astype(np.float32))
    yl, mu, sg = naive_layer_norm(d["x"], d["w"], d["b"], eps=1e-5)
    dxl, dwl, dbl = naive_layer_norm_grads(d["dy"], d["x"], d["w"], eps=1e-5)
    d.update(ln_y=yl.astype(np.float32), ln_mu=mu.astype(np.float32), ln_sigma=sg.astype(np.float32),
             ln_dx=dxl.astype(np.float32), ln_dw=dwl.astype(np.float32), ln_db=dbl.astype(np.float32))
    return d


def main():
    for name, cfg in ROW_CASES.items():
        d = make_row_case(*cfg)
        np.savez_compressed(os.path.join(HERE, f"rows_{name}.npz"), **d)
        print("rows", name, d["x"].shape)
    for name, cfg in ROPE_CASES.items():
        d = make_rope_case(*cfg)
        np.savez_compressed(os.path.join(HERE, f"rope_{name}.npz"), **d)
        print("rope", name, {k: v.shape for k, v in d.items()})
    if "--no-attention" in sys.argv:
        return
    for name, cfg in CASES.items():
        d = make_case(*cfg)
        np.savez_compressed(os.path.join(HERE, f"fa_{name}.npz"), **d)
        print(name, {k: v.shape for k, v in d.items() if hasattr(v, "shape") and v.size})


if __name__ == "__main__":
    main()
