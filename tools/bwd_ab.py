"""dev: backward time per shape under launch-shape knobs (debug_set), steady state.  usage: bwd_ab.py knob=v[,knob=v] ... -- dt:E:L:QH:KH:B:mode ..."""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
args = sys.argv[1:]
split = args.index("--")
settings, shapes = args[:split], args[split + 1:]
for c in shapes:
    dt, E, L, QH, KH, B, mode = c.split(":"); E, L, QH, KH, B = int(E), int(L), int(QH), int(KH), int(B)
    g = torch.Generator(device=dev).manual_seed(1)
    q, do = (torch.randn(B, QH, L, E, generator=g, device=dev).to(DT[dt]) for _ in range(2))
    k, v = (torch.randn(B, KH, L, E, generator=g, device=dev).to(DT[dt]) for _ in range(2))
    causal = mode == "causal"
    o, ms, ls = pkg._flash_attention(q, k, v, causal=causal)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ws = torch.empty(pkg.bwd_workspace_bytes(q, k, v, causal=causal), dtype=torch.uint8, device=dev)
    f = lambda: pkg.fa_bwd_into(dq, dk, dv, None, ws, do, o, ms, ls, q, k, v, causal=causal)
    fl = 2.5 * pkg.workmodel.attention_flops(E, L, L, QH, B, causal=causal)
    res = []
    for s in settings:
        prev = {}
        for kv in s.split(","):
            if kv == "default": continue
            kk, vv = kv.split("="); prev[kk] = pkg._lib.debug_set(kk, int(vv))
        est = fl / 0.5e9 + 20
        nwarm, n = max(10, int(0.5e6 / est)), max(10, int(0.3e6 / est))
        for _ in range(nwarm): f()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
        t = sorted(ts)[1]
        res.append(f"{s}: {t:.1f}us {fl / t / 1e6:.0f}TF")
        for kk, pv in prev.items(): pkg._lib.debug_set(kk, pv)
    print(f"{c:32s} " + " | ".join(res), flush=True)
