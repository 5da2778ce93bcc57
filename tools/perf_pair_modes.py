"""dev: pair-bias forward, and backward through both of its paths (staged: pack -> tiled kernels on the head-major scratch -> unpack;
direct: gather / scatter in the reference layout, chosen when the caller brings only the small workspace), on the reference's
pair-bias benchmark shape (benchmarks/main.jl:306-315: E 64, L 2048, H 4, B 4).  Run under tools/prof_pair_modes.sh for per-kernel times.
usage: perf_pair_modes.py [dt] [L] [H] [B]"""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = sys.argv[1:]
if os.environ.get("FWD_NW"): pkg._lib.debug_set("fwd_nw", int(os.environ["FWD_NW"]))
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a[0] if a else "bf16"]
E = 64; L, H, B = (int(x) for x in (a[1:4] if len(a) >= 4 else (2048, 4, 4)))
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g, device=dev).to(dt)
q, k, v, do = mk(B, H, L, E), mk(B, H, L, E), mk(B, H, L, E), mk(B, H, L, E)
pair = mk(B, L, L, H)
nbytes = pair.numel() * pair.element_size()
for causal in (False, True):
    o, ms, ls = torch.empty_like(q), torch.empty(B, H, L, dtype=dt, device=dev), torch.empty(B, H, L, dtype=dt, device=dev)
    tf = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, pair, causal=causal))
    tf0 = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, None, causal=causal))
    pkg.fa_fwd_into(o, ms, ls, q, k, v, pair, causal=causal)
    dq, dk, dv, dp = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(pair)
    out = {}
    ref = None
    for name, big in (("staged", True), ("direct", False)):
        ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=causal, pair=big), dtype=torch.uint8, device=dev)
        out[name] = timeit(lambda: pkg.fa_bwd_into(dq, dk, dv, dp, ws, do, o, ms, ls, q, k, v, pair, causal=causal))
        if ref is None: ref = dp.clone()
        else: assert torch.equal(ref, dp), name          # both routes write the same dpair, bit for bit
    print(f"{str(dt)[6:]} L{L} H{H} B{B} causal={int(causal)}: fwd {tf:7.1f} us (no bias {tf0:6.1f}; bias stream {nbytes / tf / 1e3:6.0f} GB/s)   "
          f"bwd staged {out['staged']:7.1f} us  direct {out['direct']:7.1f} us", flush=True)
