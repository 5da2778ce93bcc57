"""dev: backward cost per mode on the SAME non-causal problem: plain, masked mode via an all-valid key-padding mask, and the
causal run of the shape (which does ~half the work).  usage: perf_modes_bwd.py [dt:E:L:H:B ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for c in sys.argv[1:] or ["bf16:64:4096:16:4", "bf16:128:4096:16:4"]:
    dt, E, L, H, B = c.split(":"); E, L, H, B = int(E), int(L), int(H), int(B)
    q, k, v, do = (torch.randn(B, H, L, E, device=dev).to(DT[dt]) for _ in range(4))
    o = torch.empty_like(q); ms = torch.empty(B, H, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=False), dtype=torch.uint8, device=dev)
    mask = torch.ones(B, L, dtype=torch.bool, device=dev)
    res = {}
    pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False)
    res["plain"] = timeit(lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=False))
    res["masked(all valid)"] = timeit(lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=False, kpad_mask=mask))
    pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=True)
    res["causal"] = timeit(lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=True))
    print(c, " | ".join(f"{k_} {v_:.1f}" for k_, v_ in res.items()), f"| causal/plain {res['causal'] / res['plain']:.2f}", flush=True)
