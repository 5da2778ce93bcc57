"""dev: forward time vs number of kv tiles at a constant 256 workgroups (one per CU): fixed cost per workgroup + cost per tile,
for the 64-row form (knob fwd_w64=1) and the shipped default (-1).  usage: w64_scan.py [E ...]"""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
T = pkg._lib.debug_set
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for E in [int(x) for x in (sys.argv[1:] or ["64", "128"])]:
    for L in (512, 1024, 2048, 4096, 8192, 16384):
        BH = 256 * 256 // L
        H = min(BH, 16); B = BH // H
        q, k, v = (torch.randn(B, H, L, E, device=dev).to(torch.bfloat16) for _ in range(3))
        o = torch.empty_like(q); ms = torch.empty(B, H, L, dtype=torch.bfloat16, device=dev); ls = torch.empty_like(ms)
        f = lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=False)
        res = []
        for w in (0, 1):
            T("fwd_w64", w); res.append(timeit(f))
        T("fwd_w64", -1)
        fl = 4 * E * L * L * H * B
        print(f"E{E} L{L:6d} BH{BH:4d} tiles {L//64:4d}: old {res[0]:8.1f} us {fl/res[0]/1e6:7.1f} TF | w64 {res[1]:8.1f} us {fl/res[1]/1e6:7.1f} TF", flush=True)
