"""dev: the two-waves-per-SIMD forward (knob fwd_duo=1, csrc/fa_fwd_duo.hpp) against the launcher's other choice (fwd_duo=0) on the
same inputs: max |diff| relative to max |o| for o / ms / ls, NaN pattern, and the steady-state time of each.
usage: duo_check.py [dt:E:L:QH:KH:B:mode ...]      mode: plain | causal | lens (variable-length key mask) | ragged (KL = L - 37)"""
import os, sys, torch
os.environ.setdefault("NNOP_DEBUG_HOOKS", "1")          # unlock the kernel-form hook nnop_debug_set (csrc/nnop_debug.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
dev = torch.device("cuda:0")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}
T = pkg._lib.debug_set
def timeit(f, n=20, warm=3):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
MODES = [int(x) for x in os.environ.get("DUO_MODES", "0,1").split(",")]     # knob values compared (2 / 3: rows per wave forced to 64 / 32)
cfgs = sys.argv[1:] or ["bf16:64:4096:4:4:4:plain", "bf16:64:1024:4:2:2:causal", "f16:64:4096:16:4:4:lens", "f16:64:1000:4:4:2:ragged",
                        "bf16:64:2048:4:4:4:plain", "bf16:64:2048:4:4:4:causal", "bf16:64:4096:16:16:4:causal", "bf16:64:320:2:2:1:causal"]
for c in cfgs:
    dt, E, L, QH, KH, B, mode = c.split(":"); E, L, QH, KH, B = int(E), int(L), int(QH), int(KH), int(B)
    KL = L - 37 if mode == "ragged" else L
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(B, QH, L, E, generator=g, device=dev).to(DT[dt])
    k = torch.randn(B, KH, KL, E, generator=g, device=dev).to(DT[dt])
    v = torch.randn(B, KH, KL, E, generator=g, device=dev).to(DT[dt])
    mask = None; lens = None
    if mode == "lens":
        lens = torch.randint(KL // 4, KL + 1, (B,), generator=torch.Generator().manual_seed(2))
        mask = (torch.arange(KL)[None, :] < lens[:, None]).to(dev).contiguous()
    causal = mode == "causal"
    res, tm = {}, {}
    for w in MODES:  # 0: the launcher without the duo form
        T("fwd_duo", w)
        o = torch.empty_like(q); ms = torch.empty(B, QH, L, dtype=DT[dt], device=dev); ls = torch.empty_like(ms)
        f = lambda: pkg.fa_fwd_into(o, ms, ls, q, k, v, causal=causal, kpad_mask=mask)
        f(); torch.cuda.synchronize()
        res[w] = (o.float().clone(), ms.float().clone(), ls.float().clone())
        fl = pkg.workmodel.attention_flops(E, L, KL, QH, B, causal=causal, kv_lens=None if lens is None else lens.tolist())
        n = max(20, int(0.3e6 / (fl / 0.8e9 + 5)))
        tm[w] = timeit(f, n=n, warm=n)
    T("fwd_duo", -1)
    m0, m1 = MODES[0], MODES[-1]
    d = [float((torch.nan_to_num(a) - torch.nan_to_num(b)).abs().max() / torch.nan_to_num(b).abs().max().clamp_min(1e-30)) for a, b in zip(res[m1], res[m0])]
    nan = [bool(torch.isnan(a).any()) for a in res[m1]]
    nan0 = [bool(torch.isnan(a).any()) for a in res[m0]]
    times = " | ".join(f"knob {w}: {tm[w]:8.1f} us {fl/tm[w]/1e6:7.1f} TF" for w in MODES)
    print(f"{c:34s} diff o/ms/ls {d[0]:.2e} {d[1]:.2e} {d[2]:.2e} nan {nan} (first {nan0})  {times}", flush=True)
