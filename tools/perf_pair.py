import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); dev = torch.device("cuda:0")
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
E, L, H, B = 64, 2048, 4, 4           # benchmarks/main.jl:306-315
for dt in (torch.float32, torch.bfloat16):
    g = torch.Generator(device=dev).manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g, device=dev).to(dt)
    q, k, v, do = mk(B, H, L, E), mk(B, H, L, E), mk(B, H, L, E), mk(B, H, L, E)
    pair = mk(B, L, L, H)
    for causal in (False, True):
        for use_pair in (False, True):
            pr = pair if use_pair else None
            o, ms, ls = pkg._flash_attention(q, k, v, pr, causal=causal)
            tf = timeit(lambda: pkg._flash_attention(q, k, v, pr, causal=causal))
            tb = timeit(lambda: pkg.grad_flash_attention(do, o, ms, ls, q, k, v, pr, causal=causal))
            bytes_pair = pair.numel() * pair.element_size()
            print(f"{str(dt)[6:]:9s} causal={int(causal)} pair={int(use_pair)}: fwd {tf:8.1f} us  bwd {tb:8.1f} us" +
                  (f"   pair stream {bytes_pair/tf/1e3:6.1f} GB/s fwd, dpair+pair {2*bytes_pair/tb/1e3:6.1f} GB/s bwd" if use_pair else ""), flush=True)
