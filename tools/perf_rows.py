"""Bandwidth of the row-wise operators (online_softmax; later the norms) vs the HBM roofline.  Dev tool: one JSON line
per shape.  usage: python tools/perf_rows.py [softmax] [--cpu]"""
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
HBM_PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
DEV = "cuda:0"


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def line(op, shape, dt, us, nbytes, **kw):
    d = dict(op=op, shape=shape, dtype=dt, us=round(us, 2), bytes=nbytes, gbps=round(nbytes / us / 1e3, 1),
             frac_hbm=round(nbytes / us / 1e3 / HBM_PEAK, 3))
    d.update(kw)
    print(json.dumps(d), flush=True)


def softmax():
    softmax_bytes = pkg.workmodel.softmax_bytes
    for N, batch, dt in [(1024, 1024, "f32"), (4096, 16384, "f32"), (4096, 16384, "bf16"), (512, 262144, "bf16"),
                         (16384, 8192, "bf16"), (32768, 4096, "bf16"), (131072, 1024, "f32"), (4100, 16384, "f32")]:
        x = torch.randn(batch, N, device=DEV).to(DT[dt])
        y = torch.empty_like(x)
        us = timeit(lambda: pkg.online_softmax_into(y, x))
        us_t = timeit(lambda: torch.softmax(x, -1))
        line("online_softmax", f"N{N} batch{batch}", dt, us, softmax_bytes(N, batch, x.element_size()), us_torch=round(us_t, 2))
        dy = torch.randn_like(y)
        us = timeit(lambda: pkg.grad_online_softmax(dy, y))
        line("grad_online_softmax", f"N{N} batch{batch}", dt, us, softmax_bytes(N, batch, x.element_size(), bwd=True))
    if "--cpu" in sys.argv:
        from oracle.naive_softmax import naive_softmax
        xs = np.random.default_rng(0).standard_normal((4096, 4096)).astype(np.float32)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); naive_softmax(xs, dtype=np.float32); t.append(time.perf_counter() - t0)
        print(json.dumps(dict(op="online_softmax", cpu_baseline_gbps=round(softmax_bytes(4096, 4096, 4) / np.median(t) / 1e9, 2),
                              kind="port", cores=1, sample="N4096 batch4096 f32, median of 5")), flush=True)


def norms():
    norm_bytes = pkg.workmodel.norm_bytes
    for emb, n, dt in [(1024, 1024, "f32"), (4096, 16384, "bf16"), (4096, 16384, "f32"), (8192, 8192, "bf16"),
                       (768, 65536, "bf16"), (16384, 4096, "bf16"), (5120, 16384, "bf16")]:
        x = torch.randn(n, emb, device=DEV).to(DT[dt])
        dy = torch.randn(n, emb, device=DEV).to(DT[dt])
        w = torch.randn(emb, device=DEV); b = torch.randn(emb, device=DEV)
        shape = f"emb{emb} n{n}"
        nb_f, nb_b = norm_bytes(emb, n, x.element_size()), norm_bytes(emb, n, x.element_size(), bwd=True)
        us = timeit(lambda: pkg._rms_norm(x, w))
        us_t = timeit(lambda: torch.nn.functional.rms_norm(x, (emb,), w.to(x.dtype)))
        line("rms_norm", shape, dt, us, nb_f, us_torch=round(us_t, 2))
        y, rms = pkg._rms_norm(x, w)
        line("grad_rms_norm", shape, dt, timeit(lambda: pkg.grad_rms_norm(dy, rms, x, w)), nb_b)
        us = timeit(lambda: pkg._layer_norm(x, w, b))
        us_t = timeit(lambda: torch.nn.functional.layer_norm(x, (emb,), w.to(x.dtype), b.to(x.dtype)))
        line("layer_norm", shape, dt, us, nb_f, us_torch=round(us_t, 2))
        y, mu, sg = pkg._layer_norm(x, w, b)
        line("grad_layer_norm", shape, dt, timeit(lambda: pkg.grad_layer_norm(dy, mu, sg, x, w, b)), nb_b)
    if "--cpu" in sys.argv:
        from oracle.naive_norms import naive_layer_norm, naive_rms_norm
        xs = np.random.default_rng(0).standard_normal((4096, 4096)).astype(np.float32)
        ws = np.ones(4096, np.float32)
        for name, fn in (("rms_norm", lambda: naive_rms_norm(xs, ws, dtype=np.float32)),
                         ("layer_norm", lambda: naive_layer_norm(xs, ws, ws, dtype=np.float32))):
            t = []
            for _ in range(5):
                t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
            print(json.dumps(dict(op=name, cpu_baseline_gbps=round(norm_bytes(4096, 4096, 4) / np.median(t) / 1e9, 2),
                                  kind="port", cores=1, sample="emb4096 n4096 f32, median of 5")), flush=True)


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["softmax"]
    for w in which:
        globals()[w]()
