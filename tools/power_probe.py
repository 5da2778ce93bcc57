import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
B, H, L, E = 4, 4, 4096, 64
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dt in (torch.bfloat16, torch.float16):
    for name, gen in [("randn", lambda: torch.randn(B, H, L, E, device=dev)), ("zeros", lambda: torch.zeros(B, H, L, E, device=dev)),
                      ("ones*0.1", lambda: torch.full((B, H, L, E), 0.1, device=dev)), ("randn*0.01", lambda: 0.01 * torch.randn(B, H, L, E, device=dev))]:
        q, k, v = (gen().to(dt) for _ in range(3))
        o = torch.empty_like(q); ms = torch.empty(B, H, L, dtype=dt, device=dev); ls = torch.empty_like(ms)
        for sp in (0, 1):
            pkg._lib.debug_set("fwd_split", sp)
            t = timeit(lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False))
            print(f"{str(dt)[6:]:9s} {name:11s} split={sp}: {t:7.1f} us  {68719.476736 / t:7.1f} TF", flush=True)
