"""dev: per-tile cost of the forward's modes on the SAME non-causal problem: plain split-KV, plain 8-wave (knob fwd_split=0),
masked mode via an all-valid key-padding mask; and the causal run of the same shape.  usage: perf_modes.py [dt:E:L:H:B ...]"""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
T = pkg._lib.debug_set          # launch-shape knobs (csrc/tuning.hpp); -1 = automatic
T("fwd_w64", 0)
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for c in sys.argv[1:] or ["bf16:64:4096:16:4", "bf16:128:4096:16:4"]:
    dt, E, L, H, B = c.split(":"); E, L, H, B = int(E), int(L), int(H), int(B)
    q, k, v = (torch.randn(B, H, L, E, device=dev).to(DT[dt]) for _ in range(3))
    o = torch.empty_like(q); ms = torch.empty(B, H, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
    mask = torch.ones(B, L, dtype=torch.bool, device=dev)
    res = {}
    T("fwd_split", 1); res["plain split"] = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False))
    T("fwd_split", 0); res["plain 8w"] = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False))
    T("fwd_nw", 4); res["plain 4w"] = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False)); T("fwd_nw", -1)
    T("fwd_split", 1)
    res["masked(all valid) 8w"] = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False, kpad_mask=mask))
    T("fwd_nw", 4); res["masked(all valid) 4w"] = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False, kpad_mask=mask)); T("fwd_nw", -1)
    res["causal (default)"] = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=True))
    print(c, " | ".join(f"{k_} {v_:.1f}" for k_, v_ in res.items()), flush=True)
