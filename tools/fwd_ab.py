"""dev: steady-state forward time per shape under launch-shape knobs, alternated on ONE box.
usage: fwd_ab.py knob=v[,knob=v] [knob=v ...] -- dt:E:L:QH:KH:B:mode ...     (each knob set is one arm; mode: plain | causal | lens | ragged)"""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
i = sys.argv.index("--")
arms = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",") if kv) for a in sys.argv[1:i]]
for c in sys.argv[i + 1:]:
    dt, E, L, QH, KH, B, mode = c.split(":"); E, L, QH, KH, B = int(E), int(L), int(QH), int(KH), int(B)
    KL = L - 37 if mode == "ragged" else L
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(B, QH, L, E, generator=g, device=dev).to(DT[dt]); k = torch.randn(B, KH, KL, E, generator=g, device=dev).to(DT[dt])
    v = torch.randn(B, KH, KL, E, generator=g, device=dev).to(DT[dt])
    mask = None; lens = None
    if mode == "lens":
        lens = torch.randint(KL // 4, KL + 1, (B,), generator=torch.Generator().manual_seed(2))
        mask = (torch.arange(KL)[None, :] < lens[:, None]).to(dev).contiguous()
    causal = mode == "causal"
    o = torch.empty_like(q); ms = torch.empty(B, QH, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
    f = lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal, kpad_mask=mask)
    fl = pkg.workmodel.attention_flops(E, L, KL, QH, B, causal=causal, kv_lens=None if lens is None else lens.tolist())
    n = max(10, int(0.5e6 / (fl / 0.8e9 + 5)))
    res = {j: [] for j in range(len(arms))}
    for rep in range(3):
        for j, arm in enumerate(arms):
            prev = {kk: pkg._lib.debug_set(kk, vv) for kk, vv in arm.items()}
            for _ in range(n): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            res[j].append(e0.elapsed_time(e1) / n * 1e3)
            for kk, vv in prev.items(): pkg._lib.debug_set(kk, vv)
    print(c, " | ".join(f"{arms[j]} {sorted(res[j])[1]:.1f}us {fl / sorted(res[j])[1] / 1e6:.0f}TF" for j in res), flush=True)
