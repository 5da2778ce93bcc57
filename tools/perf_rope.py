"""Bandwidth of the Llama RoPE kernel vs the HBM roofline (dev tool; prints one JSON line per shape).
usage: python tools/perf_rope.py [--cpu]   (--cpu also times the numpy oracle on a bounded sample)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
rope_bytes = pkg.workmodel.rope_bytes

HBM_PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
SHAPES = [  # (name, D, L, QH, KH, B, dtype)
    ("llama8b-bf16", 128, 4096, 32, 8, 4, "bf16"),
    ("llama8b-bf16-long", 128, 32768, 32, 8, 1, "bf16"),
    ("llama8b-f32", 128, 4096, 32, 8, 4, "f32"),
    ("d64-bf16", 64, 4096, 16, 16, 8, "bf16"),
    ("ref-test-f32", 16, 1025, 5, 5, 2, "f32"),
]


def run(name, D, L, QH, KH, B, dt, iters=50):
    dev = "cuda:0"
    q = torch.randn(B, QH, L, D, device=dev).to(DT[dt])
    k = torch.randn(B, KH, L, D, device=dev).to(DT[dt])
    cos, sin = pkg.LlamaRotaryEmbedding(D)(torch.arange(L, device=dev, dtype=torch.float32).expand(B, L).contiguous())
    qo, ko = torch.empty_like(q), torch.empty_like(k)
    for _ in range(5):
        pkg.llama_rope_into(qo, ko, q, k, cos, sin)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        pkg.llama_rope_into(qo, ko, q, k, cos, sin)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    nbytes = rope_bytes(D, L, QH, KH, B, q.element_size())
    out = dict(op="llama_rope", shape=name, D=D, L=L, QH=QH, KH=KH, B=B, dtype=dt, us=round(us, 2),
               bytes=nbytes, gbps=round(nbytes / us / 1e3, 1), frac_hbm=round(nbytes / us / 1e3 / HBM_PEAK, 3))
    # the reference's schedule for comparison: copy q, k then rotate the copies in place (src/rope/llama_rope.jl:75-86)
    e0.record()
    for _ in range(iters):
        qo.copy_(q); ko.copy_(k)
        pkg.llama_rope_into(qo, ko, qo, ko, cos, sin)
    e1.record(); torch.cuda.synchronize()
    out["us_copy_then_inplace"] = round(e0.elapsed_time(e1) * 1e3 / iters, 2)
    return out


if __name__ == "__main__":
    for s in SHAPES:
        print(json.dumps(run(*s)), flush=True)
    if "--cpu" in sys.argv:
        from oracle.naive_rope import pairwise_llama_rope      # CPU baseline leg only
        D, L, QH, KH, B = 128, 4096, 32, 8, 1
        rng = np.random.default_rng(0)
        q = rng.standard_normal((B, QH, L, D)).astype(np.float32)
        k = rng.standard_normal((B, KH, L, D)).astype(np.float32)
        cs = rng.standard_normal((B, L, D)).astype(np.float32)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); pairwise_llama_rope(q, k, cs, cs, dtype=np.float32); t.append(time.perf_counter() - t0)
        nb = rope_bytes(D, L, QH, KH, B, 4)
        print(json.dumps(dict(op="llama_rope", cpu_baseline_gbps=round(nb / np.median(t) / 1e9, 2), kind="port",
                              cores=1, sample=f"D{D} L{L} QH{QH} KH{KH} B{B} f32, median of 5")), flush=True)
