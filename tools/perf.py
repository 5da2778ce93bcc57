"""Quick perf probe: python tools/perf.py [cfg ...]   cfg = dt:E:L:QH:KH:B:causal"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
cfgs = sys.argv[1:] or ["bf16:64:4096:4:4:4:0", "bf16:128:4096:8:8:2:0", "bf16:128:8192:8:8:2:1", "f32:64:4096:4:4:4:0"]
def timeit(f, n):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for c in cfgs:
    dt, E, L, QH, KH, B, causal = c.split(":")
    E, L, QH, KH, B, causal = int(E), int(L), int(QH), int(KH), int(B), bool(int(causal))
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda h: torch.randn(B, h, L, E, generator=g, device=dev).to(DT[dt])
    q, k, v, do = mk(QH), mk(KH), mk(KH), mk(QH)
    o = torch.empty_like(q); ms = torch.empty(B, QH, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=causal), dtype=torch.uint8, device=dev)
    flops = 4 * E * L * L * QH * B * (0.5 if causal else 1.0)
    n = 50 if flops < 5e11 else 10
    tf = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal), n)
    tb = timeit(lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=causal), max(n // 2, 3))
    print(f"{c:28s} fwd {tf*1e6:9.1f} us {flops/tf/1e12:7.1f} TF | bwd {tb*1e6:9.1f} us {2.5*flops/tb/1e12:7.1f} TF", flush=True)
